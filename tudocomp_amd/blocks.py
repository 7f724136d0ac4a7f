"""Block container + the one exchange step of the multi-GPU mode.

The reference has no block mode; the framing is this project's own:

    b"tdcgpu-blocks%" | u32 G | G x { u64 raw_len, u64 comp_len } | payload_0 | ... | payload_{G-1}

Every payload is byte-identical to the `--raw` lcpcomp stream of that shard.  Shards are independent, so there is no
data-path collective during compression; afterwards the per-shard streams are gathered on rank 0: an all-gather of
the sizes, then ONE group of point-to-point operations (on a GPU node: RCCL, every peer over its own xGMI link to rank 0).
"""
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
