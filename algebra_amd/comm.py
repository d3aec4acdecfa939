"""One process per GPU: how the ranks of a job find each other and get the library's RCCL communicator.

No counterpart in the reference (single process).  The launcher's environment names the job -- RANK, WORLD_SIZE,
LOCAL_RANK, MASTER_ADDR, MASTER_PORT, as `torch.distributed.run` and `bench.py --gpus N` set them -- and the only thing
the ranks must exchange before RCCL takes over is the communicator's 128-byte id: rank 0 makes it and serves it on a
TCP socket next to the master port, the others fetch it.  Nothing here imports torch.
"""

import os
import socket
import time

from . import _native

ID_PORT_OFFSET = 29          # next to MASTER_PORT, which a torch.distributed store of the same job may hold


def _serve(unique_id, address, port, world, timeout):
    with socket.socket() as server:
        server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        server.bind((address, port))
        server.listen(world)
        server.settimeout(timeout)
        for _ in range(world - 1):
            conn, _ = server.accept()
            with conn:
                conn.sendall(unique_id)


def _fetch(address, port, timeout):
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((address, port), timeout=5) as conn:
                data = b""
                while len(data) < 128:
                    chunk = conn.recv(128 - len(data))
                    if not chunk:
                        break
                    data += chunk
            if len(data) == 128:
                return data
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"no communicator id from rank 0 at {address}:{port}")
        time.sleep(0.05)


def init(ctx=None, rank=None, world=None, address=None, port=None, timeout=120):
    """`_native.Comm` of this rank (ranks and rendezvous from the environment unless given)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    address = address or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + ID_PORT_OFFSET)
    ctx = ctx or _native.default_context()
    if rank == 0:
        unique_id = _native.comm_unique_id()
        if world > 1:
            _serve(unique_id, address, port, world, timeout)
    else:
        unique_id = _fetch(address, port, timeout)
    return _native.Comm(ctx, unique_id, rank, world)
