"""CPU tests of the multi-GPU path's exchange step: world-size-2 gloo run of the shard -> gather -> container flow
(the per-shard streams come from the oracle here; on GPUs they come from the HIP path, byte-identical)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tudocomp_amd import blocks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    from oracle import oracle as O
    import tudocomp_amd as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = T.gen_english(20000 + 777 * rank, 42 + rank).tobytes()      # ragged shard sizes
        stream, _ = O.lcpcomp_huff_compress(O.escape(shard), 2, 1)
        buf = torch.zeros(len(stream) + 100, dtype=torch.uint8)             # capacity > length, like the bound()-sized buffer
        buf[:len(stream)] = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy())
        sizes, bufs = blocks.gather_streams(dist, torch, buf, len(stream), rank, world, torch.device("cpu"))
        assert sizes[rank] == len(stream)
        raw = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(raw, torch.tensor([len(shard)], dtype=torch.int64))
        if rank == 0:
            blob = blocks.pack_container([int(r.item()) for r in raw], [b.numpy().tobytes() for b in bufs])
            with open(os.path.join(tmpdir, "container.bin"), "wb") as f:
                f.write(blob)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_to_rank0_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    import tudocomp_amd as T
    blob = open(tmp_path / "container.bin", "rb").read()
    parts = blocks.unpack_container(blob)
    assert len(parts) == world
    for r, (raw_len, payload) in enumerate(parts):
        shard = T.gen_english(20000 + 777 * r, 42 + r).tobytes()
        assert raw_len == len(shard)
        want, _ = O.lcpcomp_huff_compress(O.escape(shard), 2, 1)
        assert payload == want                                   # byte-identical to the single-device stream
        assert O.unescape(O.lcpcomp_huff_decompress(payload)) == shard


def _shard4(rank):
    import tudocomp_amd as T
    if rank == 2:
        return b""                                                        # an empty shard (fewer blocks than ranks)
    return T.gen_english(15000 + 1234 * rank + (7 if rank == 3 else 0), 142 + rank).tobytes()


def _worker4(rank, world, port, tmpdir):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = _shard4(rank)
        stream = O.lcpcomp_huff_compress(O.escape(shard), 2, 1)[0] if shard else b""
        buf = torch.zeros(len(stream) + 64, dtype=torch.uint8)
        if stream:
            buf[:len(stream)] = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy())
        sizes, bufs = blocks.gather_streams(dist, torch, buf, len(stream), rank, world, torch.device("cpu"))
        assert sizes[rank] == len(stream) and len(sizes) == world
        raw = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(raw, torch.tensor([len(shard)], dtype=torch.int64))
        if rank == 0:
            blob = blocks.pack_container([int(r.item()) for r in raw], [b.numpy().tobytes() for b in bufs])
            with open(os.path.join(tmpdir, "container4.bin"), "wb") as f:
                f.write(blob)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_to_rank0_world4_with_empty_and_ragged_shards(tmp_path):
    """configs[4] rehearsed at world size 4 on the CPU (gloo): one rank has nothing to send, one has a ragged shard; the grouped
    point-to-point gather must still deliver every stream to rank 0 in rank order."""
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    parts = blocks.unpack_container(open(tmp_path / "container4.bin", "rb").read())
    assert len(parts) == world
    for r, (raw_len, payload) in enumerate(parts):
        shard = _shard4(r)
        assert raw_len == len(shard)
        if shard:
            assert payload == O.lcpcomp_huff_compress(O.escape(shard), 2, 1)[0]
            assert O.unescape(O.lcpcomp_huff_decompress(payload)) == shard
        else:
            assert payload == b""


def test_container_roundtrip_and_ranges():
    parts = [b"", b"abc", bytes(range(256))]
    blob = blocks.pack_container([0, 10, 300], parts)
    assert [(r, bytes(p)) for r, p in blocks.unpack_container(blob)] == list(zip([0, 10, 300], parts))
    assert blocks.shard_ranges(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert blocks.shard_ranges(8, 4) == [(0, 4), (4, 8)]


def _worker_shared(rank, world, port, tmpdir):
    """the exchange through ONE shared-memory container (bench.py's default for N > 1): sizes all-gathered, every rank writes its
    stream at its own offset, rank 0 the header"""
    from oracle import oracle as O
    import tudocomp_amd as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shared = None
    try:
        shard = b"" if rank == 2 else T.gen_dna(15000 + 4321 * rank, 7 + rank).tobytes()       # rank 2: an empty shard
        stream, _ = O.lcpcomp_huff_compress(O.escape(shard), 3, 1)
        name = "tdc_test_blocks_%d" % port
        if rank == 0:
            shared = blocks.SharedContainer(name, 1 << 20, create=True)
        dist.barrier()
        if rank != 0:
            shared = blocks.SharedContainer(name, 1 << 20, create=False)
        szt = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(szt, torch.tensor([len(stream)], dtype=torch.int64))
        sizes = [int(x.item()) for x in szt]
        raw = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(raw, torch.tensor([len(shard)], dtype=torch.int64))
        offs, end = blocks.payload_offsets(sizes)
        assert offs[0] == blocks.header_len(world) and end == offs[-1] + sizes[-1]
        shared.a[offs[rank]:offs[rank] + len(stream)] = np.frombuffer(stream, dtype=np.uint8)
        if rank == 0:
            shared.write_header([int(r.item()) for r in raw], sizes)
        dist.barrier()                                       # every payload is in place
        if rank == 0:
            with open(os.path.join(tmpdir, "shared_container.bin"), "wb") as f:
                f.write(shared.blob(sizes))
        dist.barrier()
    finally:
        if shared is not None:
            shared.close()
        dist.destroy_process_group()


def test_container_in_shared_memory_world3(tmp_path):
    world = 3
    port = _free_port()
    mp.spawn(_worker_shared, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as O
    import tudocomp_amd as T
    assert not os.path.exists("/dev/shm/tdc_test_blocks_%d" % port)          # rank 0 removed the segment
    parts = blocks.unpack_container(open(tmp_path / "shared_container.bin", "rb").read())
    assert len(parts) == world
    for r, (raw_len, payload) in enumerate(parts):
        shard = b"" if r == 2 else T.gen_dna(15000 + 4321 * r, 7 + r).tobytes()
        assert raw_len == len(shard)
        want, _ = O.lcpcomp_huff_compress(O.escape(shard), 3, 1)
        assert payload == want
        assert O.unescape(O.lcpcomp_huff_decompress(payload)) == shard
