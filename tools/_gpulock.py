"""Several measurement processes on one GPU box: CPU phases (data generation, the sequential oracle) run in parallel, but every
Hogwild launch must have the GPU to itself -- the engine's concurrency plan assumes all of its workgroups are resident.
`with gpu():` serialises the GPU sections of cooperating processes through an advisory file lock."""
import contextlib
import fcntl
import os

_PATH = os.environ.get("RFM_GPU_LOCK", "/tmp/rfm_gpu.lock")


@contextlib.contextmanager
def gpu():
    with open(_PATH, "a+") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)
