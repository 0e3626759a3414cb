"""Term tables of the benchmark tasks (constants transcribed from the reference's task configuration files)."""
