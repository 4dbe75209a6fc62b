"""Single-node rendezvous of the one-process-per-GPU ranks without any framework: a directory of small files.

The only things the sharded environment's ranks ever have to agree on are a 128-byte RCCL unique id (rank 0 ->
everybody), a barrier before and after a timed region, and a maximum over the ranks' clocks -- a few hundred bytes,
off the data path.  The launcher (``python -m torch.distributed.run`` in the driver's case, or any other) only has
to give every rank RANK / WORLD_SIZE and a key all ranks share (MASTER_PORT + TORCHELASTIC_RUN_ID when present).
Files are written under a temporary name and renamed, so a reader never sees half a message.
"""
from __future__ import annotations

import io
import os
import tempfile
import time
from typing import List, Optional

import numpy as np


class FileRendezvous:
    def __init__(self, rank: int, world: int, key: Optional[str] = None, root: Optional[str] = None, timeout: float = 600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = "p%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        base = root or os.environ.get("GS_RENDEZVOUS_DIR") or os.path.join(tempfile.gettempdir(), "gridstep_rdzv")
        self.base = os.path.join(base, "".join(c if c.isalnum() or c in "_-" else "_" for c in key))
        os.makedirs(self.base, exist_ok=True)
        self._seq = 0
        # Every job gets a directory of its own ("generation") under the key, announced by rank 0 in `hello`; the
        # others join it and wait for rank 0's welcome -- if that does not come (the hello was a leftover of a job that
        # died under the same key) they read `hello` again.
        hello = os.path.join(self.base, "hello")
        if self.rank == 0:
            try:
                os.unlink(hello)
            except OSError:
                pass
            gen = "gen_%d_%d" % (os.getpid(), int(time.time() * 1e6))
            self.dir = os.path.join(self.base, gen)
            os.makedirs(self.dir, exist_ok=True)
            tmp = hello + ".tmp"
            with open(tmp, "w") as f:
                f.write(f"{gen} {self.world}")
            os.replace(tmp, hello)
            for r in range(1, self.world):
                self._get(f"join.{r}")
            self._put("welcome", b"1")
        else:
            t0 = time.time()
            while True:
                try:
                    gen, w = open(hello).read().split()
                except (OSError, ValueError):
                    gen = None
                if gen is not None:
                    if int(w) != self.world:
                        raise RuntimeError(f"rendezvous {self.base}: rank 0 announced world {w}, this rank has {self.world}")
                    self.dir = os.path.join(self.base, gen)
                    if os.path.isdir(self.dir):
                        self._put(f"join.{self.rank}", b"1")
                        try:
                            self._get("welcome", timeout=5.0)
                            break
                        except TimeoutError:
                            pass
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rendezvous: rank 0 did not show up under {self.base} within {self.timeout:.0f} s")
                time.sleep(0.01)

    # -- files --------------------------------------------------------------------------------
    def _put(self, name: str, data: bytes) -> None:
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, os.path.join(self.dir, name))

    def _get(self, name: str, timeout: Optional[float] = None) -> bytes:
        path = os.path.join(self.dir, name)
        t0, limit, polls = time.time(), self.timeout if timeout is None else timeout, 0
        while True:
            try:
                with open(path, "rb") as f:
                    return f.read()
            except OSError:
                pass
            if time.time() - t0 > limit:
                raise TimeoutError(f"rendezvous: {path} did not appear within {limit:.0f} s")
            polls += 1
            if polls > 2000:                      # a barrier between ranks that are already there resolves while spinning
                time.sleep(0.001)

    def _tag(self, what: str) -> str:
        self._seq += 1
        return f"{self._seq:06d}_{what}"

    # -- collectives (every rank must call them in the same order) ---------------------------
    def all_gather_bytes(self, data: bytes) -> List[bytes]:
        tag = self._tag("ag")
        self._put(f"{tag}.{self.rank}", data)
        return [data if r == self.rank else self._get(f"{tag}.{r}") for r in range(self.world)]

    def broadcast_bytes(self, data: Optional[bytes], src: int = 0) -> bytes:
        tag = self._tag("bc")
        if self.rank == src:
            self._put(tag, data)
            # the sender may not race ahead and delete / rewrite before everybody has read: collect acknowledgements
            for r in range(self.world):
                if r != src:
                    self._get(f"{tag}.ack{r}")
            return data
        out = self._get(tag)
        self._put(f"{tag}.ack{self.rank}", b"1")
        return out

    def barrier(self) -> None:
        self.all_gather_bytes(b"1")

    def all_reduce_max(self, value: float) -> float:
        return max(float(x) for x in self.all_gather_bytes(repr(float(value)).encode()))

    def all_gather_array(self, a: np.ndarray) -> List[np.ndarray]:
        buf = io.BytesIO()
        np.save(buf, np.ascontiguousarray(a), allow_pickle=False)
        return [np.load(io.BytesIO(b), allow_pickle=False) for b in self.all_gather_bytes(buf.getvalue())]

    def close(self) -> None:
        self.barrier()
        if self.rank == 0:
            time.sleep(0.05)                     # the other ranks are reading the barrier files
            for f in os.listdir(self.dir):
                try:
                    os.unlink(os.path.join(self.dir, f))
                except OSError:
                    pass
            for path in (self.dir,):
                try:
                    os.rmdir(path)
                except OSError:
                    pass
