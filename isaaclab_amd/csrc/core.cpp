// Library-wide pieces of the C ABI: version and the thread-local error string.
#include "common.h"

#include <cstring>

namespace ilx {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

CommandList*& recording_list() {
    static thread_local CommandList* list = nullptr;
    return list;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_commands_begin(ilx_commands_t* out) {
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_commands_begin: out is NULL");
    ILX_REQUIRE(!recording_list(), ILX_EINVAL, "ilx_commands_begin: this thread is recording already");
    CommandList* list = new CommandList();
    recording_list() = list;
    *out = reinterpret_cast<ilx_commands_t>(list);
    return ILX_OK;
}

extern "C" int ilx_commands_end(void) {
    CommandList* list = recording_list();
    ILX_REQUIRE(list, ILX_EINVAL, "ilx_commands_end: this thread is not recording");
    recording_list() = nullptr;
    list->open = false;
    ILX_REQUIRE(!list->mixed, ILX_EINVAL, "ilx_commands_end: the recorded calls named more than one stream (a list is one chain)");
    return ILX_OK;
}

extern "C" int ilx_commands_count(ilx_commands_t commands, int* count) {
    ILX_REQUIRE(commands && count, ILX_EINVAL, "ilx_commands_count: NULL argument");
    *count = static_cast<int>(reinterpret_cast<CommandList*>(commands)->commands.size());
    return ILX_OK;
}

extern "C" int ilx_commands_replay(ilx_commands_t commands, void* stream) {
    ILX_REQUIRE(commands, ILX_EINVAL, "ilx_commands_replay: list is NULL");
    CommandList* list = reinterpret_cast<CommandList*>(commands);
    ILX_REQUIRE(!list->open, ILX_EINVAL, "ilx_commands_replay: the list is still being recorded");
    ILX_REQUIRE(!list->mixed, ILX_EINVAL, "ilx_commands_replay: the list names more than one stream");
    ILX_REQUIRE(!recording_list(), ILX_EINVAL, "ilx_commands_replay: this thread is recording");
    hipStream_t s = as_stream(stream);
    for (Command& c : list->commands) {
        const hipError_t e = hipLaunchKernel(c.kernel, c.grid, c.block, c.args.data(), c.shmem, s);
        if (e != hipSuccess) return check_hip(e, "ilx_commands_replay");
    }
    return ILX_OK;
}

extern "C" int ilx_commands_destroy(ilx_commands_t commands) {
    if (!commands) return ILX_OK;
    CommandList* list = reinterpret_cast<CommandList*>(commands);
    if (recording_list() == list) recording_list() = nullptr;
    delete list;
    return ILX_OK;
}

extern "C" int ilx_version(void) { return ILX_ABI_VERSION; }
extern "C" const char* ilx_last_error(void) { return ilx::error_buffer(); }
