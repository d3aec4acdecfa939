"""One process per GPU: how the ranks of a job find each other and get the library's RCCL communicator.

No counterpart in the reference (single process).  The launcher's environment names the job -- RANK, WORLD_SIZE,
LOCAL_RANK, MASTER_ADDR, MASTER_PORT, as `torch.distributed.run` and `bench.py --gpus N` set them -- and the only thing
the ranks must exchange before RCCL takes over is the communicator's 128-byte id: rank 0 makes it and serves it on a
TCP socket next to the master port, the others say who they are (job token, world size, rank) and fetch it.  Nothing
here imports torch.
"""

import os
import socket
import time

from . import _native

ID_PORT_OFFSET = 29          # next to MASTER_PORT, which a torch.distributed store of the same job may hold


def _hello(rank, world, token):
    return f"malcs {token} {world} {rank}\n".encode()


def _serve(unique_id, address, port, world, timeout, token):
    """Rank 0: hands the id to every rank 1 ... world - 1 that says who it is (job token, world size, rank); anything
    else that connects is turned away and does not use up a place.  Raises when the port cannot be bound (another job
    of the same MASTER_PORT on this host) or a rank has not come by the deadline."""
    deadline = time.monotonic() + timeout
    served = set()
    with socket.socket() as server:
        server.bind((address, port))                       # (no SO_REUSEADDR: two jobs on one port must not both succeed)
        server.listen(world)
        while len(served) < world - 1:
            left = deadline - time.monotonic()
            if left <= 0:
                missing = sorted(set(range(1, world)) - served)
                raise TimeoutError(f"ranks {missing} did not ask for the communicator id at {address}:{port}")
            server.settimeout(left)
            try:
                conn, _ = server.accept()
            except socket.timeout:
                continue
            with conn:
                conn.settimeout(5)
                try:
                    line = b""
                    while not line.endswith(b"\n") and len(line) < 64:
                        chunk = conn.recv(64 - len(line))
                        if not chunk:
                            break
                        line += chunk
                    words = line.decode("ascii", "replace").split()
                    ok = len(words) == 4 and words[0] == "malcs" and words[1] == str(token) and words[2] == str(world) and words[3].isdigit()
                    rank = int(words[3]) if ok else -1
                    if ok and 1 <= rank < world:
                        conn.sendall(unique_id)
                        served.add(rank)
                except OSError:
                    pass


def _fetch(address, port, timeout, rank, world, token):
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((address, port), timeout=5) as conn:
                conn.sendall(_hello(rank, world, token))
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
    token = os.environ.get("MALCS_JOB_TOKEN") or os.environ.get("MASTER_PORT", "29500")      # what the ranks of one job share
    if rank == 0:
        unique_id = _native.comm_unique_id()
        if world > 1:
            _serve(unique_id, address, port, world, timeout, token)
    else:
        unique_id = _fetch(address, port, timeout, rank, world, token)
    return _native.Comm(ctx, unique_id, rank, world)
