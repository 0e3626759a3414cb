"""One-rank RCCL rehearsal of the overlapped statistics all-reduce (async work objects on the collective's own stream)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaaclab_amd.distributed import OverlappedLogReducer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
red = OverlappedLogReducer(17, "cuda")
red._dist = dist  # a one-rank group would skip the collective
got = []
for k in range(50):
    buf, totals = red.begin_step(want_totals=True)
    if totals is not None:
        got.append(totals)
    buf += float(k + 1)
    red.end_step()
got += red.finish()
torch.cuda.synchronize()
assert [float(t[0]) for t in got] == [float(k + 1) for k in range(50)], got
print("overlap probe ok")
dist.destroy_process_group()
