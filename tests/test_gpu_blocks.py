"""GPU tests of the product-level block mode (pytest -m gpu): tdc_gpu_blocks_compress / _decompress of the C ABI and
`tdc --blocks`: every payload of the container is byte-identical to the oracle's stream of that block alone (own escaping,
sentinel, suffix array, Huffman table), ragged last block, blocks that contain 0x00 / 0xFF, and the device round trip."""
import os
import subprocess

import numpy as np
import pytest

import tudocomp_amd as T
from tudocomp_amd import blocks
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TDC = os.path.join(ROOT, "tudocomp_amd", "bin", "tdc")


def _data():
    rng = np.random.default_rng(11)
    return (T.gen_english(300000, 42).tobytes() + bytes(rng.integers(0, 256, size=5000, dtype=np.uint8)) + b"\x00" * 300 + b"\xff" * 200 +
            T.gen_dna(120000, 7).tobytes())


@pytest.mark.parametrize("block_size", [65536, 100003, 1 << 20])
def test_blocks_compress_payloads_equal_single_block_streams(gpu_ctx, block_size):
    data = _data()
    blob, st = T.blocks_compress(data, block_size, threshold=2, flatten=1)
    parts = blocks.unpack_container(blob)
    want_parts = [data[o:o + block_size] for o in range(0, len(data), block_size)]
    assert len(parts) == len(want_parts) == len(st)
    for k, ((raw_len, payload), part) in enumerate(zip(parts, want_parts)):
        assert raw_len == len(part)
        want, _ = O.lcpcomp_huff_compress(O.escape(part), 2, 1)
        assert bytes(payload) == want, k
        assert st[k]["out_len"] == len(want)
    assert gpu_ctx.blocks_decompress(blob) == data                       # device decoder, restrictions removed per block
    assert blocks.decompress_container(blob, lambda s: O.unescape(O.lcpcomp_huff_decompress(s))) == data


def test_blocks_two_workers_on_one_device(gpu_ctx):
    """devices = [0, 0]: the threaded branch of tdc_gpu_blocks_compress (one host thread + one context per listed device, blocks
    handed out from a shared counter) runs on the ONE GPU of the test box -- the same code that spreads blocks over eight devices.
    Seven ragged blocks; every payload must equal the oracle's stream of its block, whichever worker took it."""
    data = _data()
    block_size = 70001
    blob, st = T.blocks_compress(data, block_size, threshold=2, flatten=1, devices=[0, 0])
    parts = blocks.unpack_container(blob)
    want_parts = [data[o:o + block_size] for o in range(0, len(data), block_size)]
    assert len(parts) == len(want_parts) >= 5
    for k, ((raw_len, payload), part) in enumerate(zip(parts, want_parts)):
        assert raw_len == len(part)
        want, _ = O.lcpcomp_huff_compress(O.escape(part), 2, 1)
        assert bytes(payload) == want, k
    assert gpu_ctx.blocks_decompress(blob) == data
    # a single listed device with more blocks than devices: the library adds a second worker on its own when memory allows
    blob1, _ = T.blocks_compress(data, block_size, threshold=2, flatten=1, devices=[0])
    assert blob1 == blob


def test_arena_budget_is_reported(gpu_ctx):
    """tdc_gpu_arena_bytes / tdc_gpu_device_memory: what a shard costs and what the device has"""
    L = T._native.load()
    assert L.tdc_gpu_arena_bytes(2_000_000_001) == 112 * 2_000_000_001 + (192 << 20)
    import ctypes
    fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.tdc_gpu_device_memory(0, ctypes.byref(fr), ctypes.byref(tot)) == 0
    assert tot.value > 200e9 and fr.value <= tot.value
    assert L.tdc_gpu_device_memory(99, ctypes.byref(fr), ctypes.byref(tot)) != 0


def test_blocks_edge_cases(gpu_ctx):
    blob, st = T.blocks_compress(b"", 4096, threshold=2)
    assert blocks.unpack_container(blob) == [] and st == []
    assert gpu_ctx.blocks_decompress(blob) == b""
    blob, _ = T.blocks_compress(b"abc", 1, threshold=2)                   # one-byte blocks
    assert [r for r, _ in blocks.unpack_container(blob)] == [1, 1, 1]
    assert gpu_ctx.blocks_decompress(blob) == b"abc"
    with pytest.raises(T.TdcGpuError):
        T.blocks_compress(b"abc", 0)
    with pytest.raises(T.TdcGpuError):
        gpu_ctx.blocks_decompress(b"not a container")
    bad = bytearray(T.blocks_compress(b"hello hello hello", 8, threshold=2)[0])
    bad[-1] ^= 0x55
    try:                                                                    # a damaged payload is refused or decodes to something else, never crashes
        gpu_ctx.blocks_decompress(bytes(bad))
    except T.TdcGpuError:
        pass
    assert T.device_count() >= 1


def test_tdc_blocks_command_line_roundtrip(tmp_path):
    data = _data()
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    algo = "lcpcomp(coder=huff,threshold=2)"
    r = subprocess.run([TDC, "-a", algo, "--blocks", "131072", "-f", "-o", str(tmp_path / "c.tdc"), str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    blob = (tmp_path / "c.tdc").read_bytes()
    assert blob.startswith(algo.encode() + b"%" + blocks.MAGIC)
    parts = blocks.unpack_container(blob[len(algo) + 1:])
    assert [r_ for r_, _ in parts] == [min(131072, len(data) - o) for o in range(0, len(data), 131072)]
    assert bytes(parts[1][1]) == O.lcpcomp_huff_compress(O.escape(data[131072:262144]), 2, 1)[0]
    r = subprocess.run([TDC, "-d", "-f", "-o", str(tmp_path / "back"), str(tmp_path / "c.tdc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "back").read_bytes() == data


def test_raw_entry_point_accepts_more_than_2_pow_30_bytes_of_plain_text(gpu_ctx):
    """the raw entry point used to stop at 2^30 input bytes (worst-case escaping); the escapes are now counted on the device"""
    n = (1 << 30) + 4096
    data = T.gen_english(n, 43)
    out, st = gpu_ctx.lcpcomp_compress_raw(data, 2, 1)
    assert st["n"] == n + 1 and len(out) == st["out_len"]
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
    assert O.lcpcomp_huff_decompress(out) == text.tobytes()


def test_kept_stream_is_fetched_into_registered_shared_memory(gpu_ctx, tmp_path):
    """tdc_gpu_lcpcomp_compress_keep / tdc_gpu_stream_fetch / tdc_gpu_host_register: the one-process-per-GPU block mode downloads
    every rank's stream to its offset of a container in shared memory (bench.py N > 1)"""
    import numpy as np
    from oracle import oracle as O
    data = T.gen_english(3_000_000, 17).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    ta = np.frombuffer(text, dtype=np.uint8)
    ln, st = gpu_ctx.lcpcomp_compress_keep(ta, len(ta), 2, 1)
    assert ln == len(want) and st["ms_d2h"] == 0
    shared = blocks.SharedContainer("tdc_test_keep_%d" % os.getpid(), 4 << 20, create=True)
    try:
        assert shared.register(T.host_register)
        off = blocks.header_len(1)
        assert gpu_ctx.stream_fetch(shared.a[off:off + ln]) == ln
        shared.write_header([len(data)], [ln])
        assert blocks.unpack_container(shared.blob([ln])) == [(len(data), want)]
        again = np.zeros(ln + 5, dtype=np.uint8)                 # the stream may be fetched again, into pageable memory too
        assert gpu_ctx.stream_fetch(again) == ln and again[:ln].tobytes() == want
        with pytest.raises(T.TdcGpuError):
            gpu_ctx.stream_fetch(np.zeros(ln - 1, dtype=np.uint8))          # too small
        gpu_ctx.suffix_array(O.escape(b"abracadabra"))                     # any other call: the kept stream is gone
        with pytest.raises(T.TdcGpuError):
            gpu_ctx.stream_fetch(again)
    finally:
        shared.close(T.host_unregister)


def test_blocks_container_of_several_megabytes(gpu_ctx):
    """above 1 MiB the payloads are copied into the container by several host threads side by side"""
    data = T.gen_english(6_300_000, 77).tobytes()
    block_size = 1_500_000
    blob, _ = T.blocks_compress(data, block_size, threshold=2, flatten=1)
    parts = blocks.unpack_container(blob)
    assert [r for r, _ in parts] == [1_500_000] * 4 + [300_000] and len(blob) > (1 << 20)
    for k, (raw_len, payload) in enumerate(parts):
        single, _ = gpu_ctx.lcpcomp_compress_raw(data[k * block_size:(k + 1) * block_size], 2, 1)
        assert bytes(payload) == single, k
    assert gpu_ctx.blocks_decompress(blob) == data


def test_bench_launches_its_own_ranks_two_rank_rehearsal(gpu_ctx):
    """`python bench.py --gpus 2` WITHOUT torchrun's environment: the parent runs the CPU baseline and starts the two ranks as a child
    process (torch.distributed.run); both ranks share this box's one GPU, the collectives run on the gloo backend.  The line must carry
    both exchange variants (shared host memory = value, the gather to rank 0 beside it), the roofline object and the CPU baseline."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TDC_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", str(1 << 24), "--steps", "2", "--warmup", "1",
                        "--cpu-sample", str(1 << 22), "--cpu-multi-sample", str(1 << 20), "--no-extra"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["world_size"] == 2
    assert j["exchange"] == "shared host memory" and j["container_ok"] is True
    assert j["rccl_gather"]["container_ok"] is True and j["rccl_gather"]["value"] > 0
    assert j["roofline"]["achieved"] > 0 and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["bit_exact_vs_gpu_on_sample"] is True
    assert j["value"] > 0 and j["scaling"] == "weak"


def test_kept_stream_is_fetched_to_device_memory(gpu_ctx):
    """tdc_gpu_stream_fetch_dev: the kept stream as the send buffer of the RCCL gather (device-to-device).  Device memory comes from
    the HIP runtime directly (ctypes), so the test does not depend on torch's device initialisation."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    data = T.gen_english(2_000_000, 5).tobytes()
    text = O.escape(data)
    want, _ = O.lcpcomp_huff_compress(text, 2, 1)
    ta = np.frombuffer(text, dtype=np.uint8)
    ln, _ = gpu_ctx.lcpcomp_compress_keep(ta, len(ta), 2, 1)
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), ln + 64) == 0
    try:
        assert gpu_ctx.stream_fetch_dev(d.value, ln + 64) == ln
        back = np.zeros(ln, dtype=np.uint8)
        assert hip.hipMemcpy(back.ctypes.data_as(ctypes.c_void_p), d, ln, 2) == 0       # hipMemcpyDeviceToHost
        assert back.tobytes() == want
        with pytest.raises(T.TdcGpuError):
            gpu_ctx.stream_fetch_dev(d.value, ln - 1)
    finally:
        hip.hipFree(d)
