"""Block container + the one exchange step of the multi-GPU mode.

The reference has no block mode; the framing is this project's own:

    b"tdcgpu-blocks%" | u32 G | G x { u64 raw_len, u64 comp_len } | payload_0 | ... | payload_{G-1}

Every payload is byte-identical to the `--raw` lcpcomp stream of that shard.  Shards are independent, so there is no
data-path collective during compression.  Afterwards the container is put together in the node's host memory, one of two ways:
  * SharedContainer (default where POSIX shared memory can be mapped): all ranks map ONE segment; after an all-gather of the
    stream sizes every rank downloads its own stream straight to its offset (tdc_gpu_stream_fetch) -- eight shards travel over
    eight host links at once, nothing passes through rank 0's GPU;
  * gather_streams: all-gather of the sizes, then ONE group of point-to-point operations (on a GPU node: RCCL, every peer over its
    own xGMI link to rank 0), and rank 0 downloads the whole container over its one host link.
"""
import os
import struct

MAGIC = b"tdcgpu-blocks%"


def pack_container(raw_lens, payloads):
    assert len(raw_lens) == len(payloads)
    head = MAGIC + struct.pack("<I", len(payloads))
    for r, p in zip(raw_lens, payloads):
        head += struct.pack("<QQ", int(r), len(p))
    return head + b"".join(bytes(p) for p in payloads)


def unpack_container(blob):
    if not blob.startswith(MAGIC):
        raise ValueError("not a tdcgpu-blocks container")
    off = len(MAGIC)
    (g,) = struct.unpack_from("<I", blob, off)
    off += 4
    dirs = [struct.unpack_from("<QQ", blob, off + 16 * i) for i in range(g)]
    off += 16 * g
    out = []
    for raw_len, comp_len in dirs:
        out.append((raw_len, blob[off:off + comp_len]))
        off += comp_len
    if off != len(blob):
        raise ValueError("trailing bytes in container")
    return out


def decompress_container(blob, decode_block):
    """inverse of the block mode: decode_block(payload) -> the block's raw bytes (restrictions already removed)"""
    out = []
    for raw_len, payload in unpack_container(blob):
        part = decode_block(bytes(payload))
        if len(part) != raw_len:
            raise ValueError("block length mismatch")
        out.append(part)
    return b"".join(out)


def shard_ranges(total, shard):
    """Contiguous byte ranges [k*shard, (k+1)*shard); the last one may be shorter."""
    return [(o, min(o + shard, total)) for o in range(0, total, shard)]


def header_len(world):
    return len(MAGIC) + 4 + 16 * world


def payload_offsets(sizes):
    """byte offset of every rank's payload in the container (tightly packed behind the header)"""
    offs, o = [], header_len(len(sizes))
    for s in sizes:
        offs.append(o)
        o += int(s)
    return offs, o


class SharedContainer:
    """The block container of one node in POSIX shared memory.  Rank 0 creates the segment, the others attach after a barrier; every
    rank may page-lock its mapping (`register`) so that downloads into it run at PCIe rate.  `a` is the numpy uint8 view."""

    def __init__(self, name, capacity, create):
        import numpy as np
        self.path = os.path.join("/dev/shm", name)
        self.capacity = int(capacity)
        self.created = bool(create)
        self.registered = False
        if create:
            with open(self.path, "wb") as f:
                f.truncate(self.capacity)
        self.a = np.memmap(self.path, dtype=np.uint8, mode="r+", shape=(self.capacity,))

    def register(self, host_register):
        self.registered = bool(host_register(self.a))
        return self.registered

    def write_header(self, raw_lens, sizes):
        head = MAGIC + struct.pack("<I", len(sizes))
        for r, c in zip(raw_lens, sizes):
            head += struct.pack("<QQ", int(r), int(c))
        self.a[:len(head)] = memoryview(head)

    def blob(self, sizes):
        """the container as bytes (tests; a writer would hand `a[:end]` to write())"""
        _, end = payload_offsets(sizes)
        return bytes(self.a[:end])

    def unlink(self):
        """remove the name from /dev/shm (rank 0, once every rank has mapped the segment): the mappings stay valid, and nothing is
        left behind -- holding host memory until reboot -- if a rank is killed before close()"""
        if self.created:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.created = False

    def close(self, host_unregister=None):
        if self.registered and host_unregister is not None:
            host_unregister(self.a)
            self.registered = False
        self.a = None
        if self.created:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def gather_streams(dist, torch, stream, length, rank, world, device):
    """Variable-size gather of per-shard streams to rank 0.

    `stream` is a uint8 tensor on `device` whose first `length` bytes are this rank's compressed shard.
    Returns (sizes, bufs): sizes on every rank, bufs (list of uint8 tensors, one per rank) on rank 0 else None."""
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([length], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    bufs = None
    # one GROUP of point-to-point operations (ncclGroupStart ... ncclGroupEnd under RCCL): rank 0 posts all its receives at
    # once, so the peers' streams arrive over their own xGMI links concurrently instead of one after the other
    if rank == 0:
        bufs = [stream[:length]] + [torch.empty(sizes[r], dtype=torch.uint8, device=device) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(1, world) if sizes[r]]
    else:
        ops = [dist.P2POp(dist.isend, stream[:length].contiguous(), 0)] if length else []
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return sizes, bufs
