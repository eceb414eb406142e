# -*- coding: utf-8 -*-
"""Single-command multi-GPU launch: the reference starts its DDP ranks from ONE `python run.py -c cfg`
(Lightning's `devices: N, strategy: ddp`, /root/reference/run.py:191-218, configs/stage2_vq.yaml:209-212).
Here the parent process -- which must not have touched the GPU yet -- starts N ranks as CHILD processes through
`python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1) and exits with their return code.
Nothing is ever re-exec'ed: a process that has initialised HIP must not be replaced by another program."""
import os
import socket
import subprocess
import sys


def under_launcher() -> bool:
    """True inside a rank started by torchrun / torch.distributed.run (or any launcher that exports RANK)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def visible_gpus() -> int:
    """Number of GPUs without initialising HIP (torch.cuda.device_count() only counts devices on this image)."""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def spawn_ranks(n: int, script: str, argv, env=None, port=None) -> int:
    """Run `python script argv...` as n ranks of one node; returns the launcher's exit code (non-zero if any rank failed)."""
    n = int(n)
    if n < 1:
        raise ValueError(f"spawn_ranks: n = {n}")
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port()), script] + list(argv)
    return subprocess.call(cmd, env=e)
