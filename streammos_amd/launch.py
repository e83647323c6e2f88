"""One process per GPU, started by the program itself.

The reference starts its multi-GPU runs with ``python -m torch.distributed.launch --nproc_per_node=N``
(README.md:97, val_StreamMOS.py:205-218).  ``bench.py --gpus N`` and ``python -m streammos_amd.run_sequence`` with
several sequences do the same thing for themselves: a parent that has NOT touched the GPU (no HIP call, no
``torch.cuda.is_available()``) starts ``torch.distributed.run`` as a child process with N ranks, relays the children's
output and exits with the child's code.  Nothing is exec'ed over a process that initialised the GPU, and the parent
never initialises it.
"""
import os
import socket
import subprocess
import sys


def under_launcher():
    """True inside a rank started by torch.distributed.run / launch (the env contract of val_StreamMOS.py:205-211)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks, script_args, module=None, script=None, timeout=None):
    """Run ``script`` (a path) or ``module`` (``-m`` name) with ``script_args`` as n_ranks ranks on this node.
    Returns the child's exit code; stdout / stderr of the ranks pass through unchanged (rank 0 prints the one JSON
    line).  The rendezvous is 127.0.0.1 on a free port: the container hostname may not resolve."""
    assert (module is None) != (script is None)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(n_ranks, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", module] if module else [script]) + list(script_args)
    try:
        return subprocess.run(cmd, env=env, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        return 124
