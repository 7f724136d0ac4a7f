#!/usr/bin/env python3
"""bench.py -- input MB/s of lcpcomp(coder=huff) on MI355X, END TO END (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic text: pinned host text -> H2D -> suffix array ->
ISA/Phi/PLCP -> ArraysComp factorization -> flatten -> host Huffman table -> bit-pack -> D2H -> compressed bytes in pinned
host memory.  Both transfers are INSIDE the timed region; `value` = input bytes / wall time.

  N = 1 : the configuration the metric is quoted on: 2*10^9 B English-like text (SURVEY.md 8d generator, seed 42),
          threshold 2, flatten 1, through the product's entry point tdc_gpu_lcpcomp_compress_into.
          Extra keys (never `value`): "hbm_resident" = the same steps without the two transfers (device time of the
          same calls), "configs1_256MiB" = BASELINE.json configs[1] (256 MiB) through the same entry point,
          "configs2_dna_1e9" = BASELINE.json configs[2] (10^9 B DNA, LCPCompressor + ArithmeticCoder, threshold 5) through
          the same entry point, its stream checked against the CPU oracle on a 32 MiB DNA sample,
          "configs3_lz78" = BASELINE.json configs[3] on a 32 MiB sample (host-bound: the LZ78 parse runs on the host),
          "decompress" = SURVEY 8f #2: the stream of the timed steps back to the text (device parse + references), compared with the input.
          "stream_matches_golden" (headline and extras): size + SHA-256 of the device stream against the ORACLE's stream of the
          same full-size text (tests/golden/oracle_fullsize.json).
  N > 1 : BASELINE.json configs[4]: one process per GPU (torch.distributed, backend nccl = RCCL); every rank compresses
          its own 2*10^9 B shard (seed 42 + rank: weak scaling) exactly as in the N = 1 case (pinned host text, upload
          overlapped with the first partition level) and keeps the stream on its GPU.  BOTH ways of putting the block
          container together are timed, K steps each, in the same run:
            "shared host memory" -- all-gather of the stream sizes, every rank downloads its stream to its offset of ONE
                container that all ranks map (POSIX shared memory, page-locked): eight shards over eight host links at once;
            "rccl gather to rank 0" -- the collective north_star names: all-gather of the sizes, ONE group of point-to-point
                operations over xGMI (ncclGroupStart .. ncclGroupEnd), rank 0 downloads the whole container over its one host link.
          `value` is the shared-host-memory run (the faster design; "exchange" says so), the RCCL gather is reported beside it
          as "rccl_gather": {"ms_per_step", "value"}; where the shared segment cannot be set up the RCCL gather is `value`.
          value = bytes of all ranks / max-over-ranks time.
          Started WITHOUT torchrun's environment (`python bench.py --gpus N`), the process launches its own N ranks: it runs the
          CPU baseline, then starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process (it has
          not touched the GPU and never exec()s), relays rank 0's JSON line and the exit code.
          TDC_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (collectives on CPU tensors).

After the timed steps ONE more step runs with per-kernel HIP-event timing switched on (untimed) -- the roofline object and
the per-kernel table come from it.  Prints ONE JSON line (rank 0).  PyTorch is used for torch.distributed and the
multi-GPU staging buffers only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STAT_KEYS = ("out_len", "factors", "maxlcp", "num_flattened", "sa_rounds", "levels", "mis_rounds", "flatten_rounds", "pushes",
             "arena_bytes", "sa_sorted_elems", "sa_init_syms", "small_levels", "purges", "window_pass", "window_lcut",
             "sa_key_words", "sa_text_rounds", "sa_mode", "sa_overlapped")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=2_000_000_000, help="input bytes per GPU (default 2*10^9: the metric's text / one shard of configs[4])")
    ap.add_argument("--threshold", type=int, default=2)
    ap.add_argument("--gen", default="english", choices=["english", "dna"])
    ap.add_argument("--cpu-sample", type=int, default=1 << 26, help="bytes of the workload the single-core CPU baseline is timed on")
    ap.add_argument("--cpu-multi-sample", type=int, default=1 << 25, help="bytes per process of the multi-core CPU figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra figures (256 MiB English, 10^9 B DNA)")
    ap.add_argument("--dna-sample", type=int, default=1 << 25, help="bytes of DNA text the oracle compresses for the configs[2] check")
    ap.add_argument("--cpu-result", default=None, help=argparse.SUPPRESS)      # (N > 1 self-launch: the parent's CPU baseline, a JSON file)
    return ap.parse_args()


# ---- CPU baseline: runs BEFORE this process touches the GPU (child processes are plain CPU programs) -------------------------
_CPU_CHILD = r"""
import sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O
gen, seed, m, thr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
coder = sys.argv[5] if len(sys.argv) > 5 else "huff"
text = (T.gen_english if gen == "english" else T.gen_dna)(m, seed)
sample = np.concatenate([text, np.zeros(1, dtype=np.uint8)])
t0 = time.perf_counter()
out, st = (O.lcpcomp_arith_compress if coder == "arith" else O.lcpcomp_huff_compress)(sample, thr, 1)
dt = time.perf_counter() - t0
print("%%.6f %%d %%s" %% (dt, len(out), hashlib.sha256(out).hexdigest()))
"""


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = "gcc -O2 (oracle/Makefile)"
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.strip().startswith("CFLAGS"):
                flags = "gcc " + line.split("=", 1)[1].strip()
                break
    except OSError:
        pass
    return model, flags


def cpu_baseline(args, seed):
    """The oracle (bit-exact CPU port of the reference path).  (i) one core on the first --cpu-sample bytes of the workload
    text (the generator's output for a shorter length is a prefix of the longer one); (ii) min(cores, 16) processes on
    independent shards (seed + k) of --cpu-multi-sample bytes each."""
    code = _CPU_CHILD % ROOT
    m1 = min(args.cpu_sample, args.size)

    def run(gen_seed, m):
        return subprocess.Popen([sys.executable, "-c", code, args.gen, str(gen_seed), str(m), str(args.threshold)],
                                stdout=subprocess.PIPE, text=True)

    p = run(seed, m1)
    dt1, len1, sha1 = p.communicate()[0].split()
    if p.returncode:
        raise RuntimeError("cpu baseline child failed")
    dt1 = float(dt1)
    model, flags = cpu_info()
    res = {"value": round(m1 / 1e6 / dt1, 3), "unit": "MB/s", "cores": 1, "kind": "port",
           "sample": "first %d bytes of the workload text, lcpcomp(coder=huff,threshold=%d,flatten=1), %.1f s" % (m1, args.threshold, dt1),
           "cpu_model": model, "compiler_flags": flags, "host_cores": os.cpu_count()}
    procs = max(1, min(os.cpu_count() or 1, 16))
    m2 = min(args.cpu_multi_sample, args.size)
    t0 = time.perf_counter()
    ps = [run(seed + 1000 + k, m2) for k in range(procs)]
    ok = all(q.communicate()[0] and q.returncode == 0 for q in ps)
    dtm = time.perf_counter() - t0
    if ok:
        res["multi_core"] = {"value": round(procs * m2 / 1e6 / dtm, 3), "unit": "MB/s", "cores": procs,
                             "sample": "%d processes x %d-byte independent shards (seeds %d..), wall %.1f s incl. process start + text generation"
                                       % (procs, m2, seed + 1000, dtm)}
    dna_ref = None
    if not args.no_extra and args.gen == "english" and args.size > (1 << 30):
        # configs[2] check: the oracle's LCPCompressor<ArithmeticCoder> stream of a DNA sample (threshold 5), compared with the GPU's below
        md = args.dna_sample
        q = subprocess.Popen([sys.executable, "-c", code, "dna", "7", str(md), "5", "arith"], stdout=subprocess.PIPE, text=True)
        o = q.communicate()[0].split()
        if q.returncode == 0 and len(o) == 3:
            dna_ref = (md, int(o[1]), o[2], float(o[0]))
    return res, (m1, int(len1), sha1), dna_ref


def golden_entry(name):
    """size + SHA-256 of the ORACLE's stream for a full-size configuration (tests/golden/oracle_fullsize.json, written by
    tests/make_fullsize_golden.py): data, not code -- the oracle itself is not touched here"""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json"))).get(name)
    except (OSError, ValueError):
        return None


def matches_golden(name, stream_view):
    """True / False: the stream equals the oracle's byte for byte (size + SHA-256); None: no committed entry for this configuration"""
    import hashlib
    g = golden_entry(name)
    if g is None:
        return None
    return bool(len(stream_view) == g["size"] and hashlib.sha256(stream_view).hexdigest() == g["sha256"])


def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun's environment: this process -- which has NOT touched the GPU -- runs the CPU baseline and
    starts the N ranks as a fresh child process (never exec), then relays rank 0's JSON line and the child's exit code."""
    import socket
    import tempfile
    cpu_file = None
    if not args.no_cpu_baseline:
        seed0 = 42 if args.gen == "english" else 7
        cpu_res, cpu_ref, dna_ref = cpu_baseline(args, seed0)
        fd, cpu_file = tempfile.mkstemp(prefix="tdc_bench_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump({"cpu_res": cpu_res, "cpu_ref": cpu_ref, "dna_ref": dna_ref}, f)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--size", str(args.size),
           "--threshold", str(args.threshold), "--gen", args.gen, "--no-cpu-baseline"]
    if args.no_extra:
        cmd.append("--no-extra")
    if cpu_file:
        cmd += ["--cpu-result", cpu_file]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    finally:
        if cpu_file:
            try:
                os.unlink(cpu_file)
            except OSError:
                pass
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.exit(p.returncode if p.returncode else (0 if line is not None else 1))


def bind_to_gpu_numa_node(torch, dev_index):
    """Run this process (and the threads it starts) on the CPUs of the NUMA node the GPU hangs on, so that the pinned text / stream
    buffers are first touched there and the transfers do not cross the socket link: with one process per GPU nobody else does it.
    Best effort: returns the node or None (no topology information, or none of the node's CPUs are allowed to this process)."""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                               # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    N = args.size
    n = N + 1                                            # generators emit no 0x00 / 0xFF: text = data + sentinel
    if n >= 0x7FFFFFFF:
        raise SystemExit("--size must stay below 2^31 - 2 (32-bit len_t of the reference)")
    seed0 = 42 if args.gen == "english" else 7
    seed = seed0 + rank

    cpu_res = cpu_ref = dna_ref = None
    if rank == 0:
        if args.cpu_result:                              # N > 1 self-launch: measured by the parent process
            try:
                j = json.load(open(args.cpu_result))
                cpu_res, cpu_ref, dna_ref = j["cpu_res"], j["cpu_ref"], j["dna_ref"]
            except (OSError, ValueError, KeyError):
                cpu_res = None
        elif not args.no_cpu_baseline:
            cpu_res, cpu_ref, dna_ref = cpu_baseline(args, seed0)     # before any GPU initialisation in this process

    import hashlib
    import numpy as np
    import torch
    import tudocomp_amd as T
    from tudocomp_amd.blocks import gather_streams, MAGIC, SharedContainer, payload_offsets, header_len, unpack_container

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    backend = os.environ.get("TDC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()   # (gloo rehearsal: ranks may share a GPU)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = device if backend == "nccl" else torch.device("cpu")     # where the tensors of the collectives live
    numa_node = bind_to_gpu_numa_node(torch, dev_index) if not os.environ.get("TDC_BENCH_NO_NUMA") else None
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    # ---- synthetic input in pinned host memory ---------------------------------------------------------------------------
    gen = T.gen_english if args.gen == "english" else T.gen_dna
    h_text = T.PinnedBuffer(n)
    gen(N, seed, out=h_text.a)
    h_text.a[N] = 0
    ctx = T.Context(dev_index)
    ctx.reserve(n)
    out_cap = N + (1 << 20)                               # the stream of these texts is < 0.5 N; a larger one fails loudly
    h_out = T.PinnedBuffer(out_cap) if world == 1 else None

    def all_ok(ok):                                       # the same collective on every rank, whatever happened before it
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    shared = None
    gather_ok = False
    d_out = h_stage = h_container = None
    cont_cap = world * (out_cap // 2) + 4096
    if world > 1:
        # (1) the block container of the node: one shared-memory segment that every rank maps and page-locks.  Every rank runs the
        #     same collectives whatever fails locally: create (rank 0) | barrier | attach (others) | all-reduce of the outcome.
        name = "tdc_blocks_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run"))
        ok = True
        if rank == 0:
            try:
                shared = SharedContainer(name, cont_cap, create=True)
            except Exception:
                shared, ok = None, False
        dist.barrier()
        if rank != 0:
            try:
                shared = SharedContainer(name, cont_cap, create=False)
            except Exception:
                shared, ok = None, False
        if shared is not None:
            try:
                ok = bool(shared.register(T.host_register)) and ok
            except Exception:
                ok = False
        shared_ok = all_ok(ok)                            # (also: every rank that will map the segment has mapped it)
        if shared is not None and rank == 0:
            shared.unlink()                               # the mappings stay valid; nothing is left behind in /dev/shm if a rank dies
        if not shared_ok and shared is not None:
            shared.close(T.host_unregister)
            shared = None
        # (2) staging for the RCCL gather to rank 0: the kept stream is copied into a torch tensor the collective can send
        ok = True
        try:
            if backend == "nccl":
                d_out = torch.empty(out_cap // 2, dtype=torch.uint8, device=device)
            else:                                         # gloo rehearsal: the "gather" runs on CPU tensors
                h_stage = torch.empty(out_cap // 2, dtype=torch.uint8)
            if rank == 0:
                h_container = torch.empty(cont_cap, dtype=torch.uint8).pin_memory()
        except Exception:
            ok = False
        gather_ok = all_ok(ok) and not os.environ.get("TDC_BENCH_NO_RCCL_GATHER")
        if os.environ.get("TDC_BENCH_RCCL_GATHER") and gather_ok and shared is not None:     # (the gather as `value`: tests)
            shared.close(T.host_unregister)
            shared = None
        if shared is None and not gather_ok:
            raise SystemExit("neither the shared container nor the gather buffers could be set up")

    def step_single():
        out_len, st = ctx.lcpcomp_compress_into(h_text, n, h_out, args.threshold, 1)
        return out_len, st, None

    def step_shared():
        out_len, st = ctx.lcpcomp_compress_keep(h_text, n, args.threshold, 1)       # as N = 1, the stream stays in HBM
        szt = [torch.zeros(1, dtype=torch.int64, device=coll_device) for _ in range(world)]
        dist.all_gather(szt, torch.tensor([out_len], dtype=torch.int64, device=coll_device))
        sizes = [int(x.item()) for x in szt]
        offs, end = payload_offsets(sizes)
        if end > shared.capacity:
            raise SystemExit("block container too small")
        ctx.stream_fetch(shared.a[offs[rank]:offs[rank] + out_len])                  # D2H to this rank's place in the container
        if rank == 0:
            shared.write_header([N] * world, sizes)
        return out_len, st, sizes

    def step_gather():
        out_len, st = ctx.lcpcomp_compress_keep(h_text, n, args.threshold, 1)       # as N = 1, the stream stays in HBM
        if backend == "nccl":
            ctx.stream_fetch_dev(d_out.data_ptr(), d_out.numel())                    # device-to-device: the send buffer of the gather
            sizes, bufs = gather_streams(dist, torch, d_out, out_len, rank, world, device)
        else:
            ctx.stream_fetch(h_stage.numpy())
            sizes, bufs = gather_streams(dist, torch, h_stage, out_len, rank, world, torch.device("cpu"))
        if rank == 0:                                     # block container -> (pinned) host memory of rank 0
            offs, end = payload_offsets(sizes)
            if end > h_container.numel():
                raise SystemExit("block container too small")
            for r, b in enumerate(bufs):
                h_container[offs[r]:offs[r] + b.numel()].copy_(b, non_blocking=True)
            if backend == "nccl":
                torch.cuda.synchronize()
            head = bytearray(MAGIC) + len(sizes).to_bytes(4, "little")
            for c in sizes:
                head += int(N).to_bytes(8, "little") + int(c).to_bytes(8, "little")
            h_container[:len(head)] = torch.frombuffer(head, dtype=torch.uint8)
        return out_len, st, sizes

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        """W untimed + K timed steps between barrier + synchronise on both sides, max over ranks"""
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        dev = []
        res = None
        for _ in range(args.steps):
            res = step()
            dev.append((res[1]["ms_total"], res[1]["ms_h2d"], res[1]["ms_d2h"]))
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, dev, res

    def check_container(out_len, sizes, blob_of_rank0):
        """untimed: every rank downloads its kept stream once more into private memory; rank 0 compares the hashes with what it parses
        out of the container of the last step"""
        mine = np.empty(out_len, dtype=np.uint8)
        ctx.stream_fetch(mine)
        hs = hashlib.sha256(mine).digest()
        hv = [torch.zeros(32, dtype=torch.uint8, device=coll_device) for _ in range(world)]
        dist.all_gather(hv, torch.frombuffer(bytearray(hs), dtype=torch.uint8).to(coll_device))
        if rank != 0:
            return None
        parts = unpack_container(blob_of_rank0())
        return bool(len(parts) == world and all(
            int(parts[r][0]) == N and hashlib.sha256(parts[r][1]).digest() == bytes(hv[r].cpu().numpy()) for r in range(world)))

    rccl = None
    container_ok = None
    if world == 1:
        step, exchange = step_single, None
    else:
        step, exchange = (step_shared, "shared host memory") if shared is not None else (step_gather, "rccl gather to rank 0")
    dt, dev_ms, (out_len, st, sizes) = timed(step)
    ranks_seen = world
    if world > 1:
        bitmap = torch.tensor([1 << rank], dtype=torch.int64, device=coll_device)
        dist.all_reduce(bitmap, op=dist.ReduceOp.SUM)
        ranks_seen = bin(int(bitmap.item())).count("1")
        if shared is not None:
            container_ok = check_container(out_len, sizes, lambda: shared.blob(sizes))
            if gather_ok:                                 # the collective north_star names, timed in the same run
                dt_g, _, (ol_g, _, sizes_g) = timed(step_gather)
                ok_g = check_container(ol_g, sizes_g, lambda: bytes(h_container[:payload_offsets(sizes_g)[1]].numpy()))
                rccl = {"exchange": "rccl gather to rank 0" if backend == "nccl" else "gather to rank 0 on the %s backend (rehearsal)" % backend,
                        "ms_per_step": round(dt_g / args.steps * 1e3, 3), "value": round(world * N / 1e6 / (dt_g / args.steps), 2), "unit": "MB/s",
                        "steps": args.steps, "warmup": args.warmup, "container_ok": ok_g,
                        "note": "same shards and steps; all-gather of the stream sizes, one group of point-to-point operations to rank 0, rank 0 downloads the container over its one host link"}
        else:
            container_ok = check_container(out_len, sizes, lambda: bytes(h_container[:payload_offsets(sizes)[1]].numpy()))

    # ---- one more step with per-kernel timing (untimed): roofline + kernel table ----------------------------------------------
    ctx.set_profiling(True)
    ctx.reset_profile()
    _, st_prof, _ = step()
    prof = ctx.kernel_profile()
    ctx.set_profiling(False)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * N / 1e6 / (dt / args.steps)
        DOMINANT = max(prof, key=lambda name: prof[name]["ms"]) if prof else None
        k = prof.get(DOMINANT, {"ms": 0.0, "launches": 0, "bytes": 0})
        roof = None
        if k["launches"]:
            achieved = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    ent = j.get("kernels", {}).get(DOMINANT) if j.get("workload_bytes") == N else None
                    if ent is None and j.get("workload_bytes") == N and j.get("kernel") == DOMINANT:
                        ent = j
                    if ent is not None:
                        traffic = ent.get("hbm_bytes_per_launch")
                        traffic_src = "static: %s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this workload, not measured in this run)" % j.get("source", "profiles/pmc_summary.json")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": DOMINANT, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launches_per_step": k["launches"],
                    "avg_launch_ms": round(k["ms"] / k["launches"], 4),
                    "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"]),
                    "share_of_device_time": round(k["ms"] / st_prof["ms_total"], 3) if st_prof["ms_total"] else None,
                    "timing": "HIP events on the library's stream around every launch, one profiled step after the timed region"}
        c_ratio = out_len / N
        z = st["factors"]
        b_alg = 44 + c_ratio + 48 * z / N
        tot = sum(d[0] for d in dev_ms) / len(dev_ms)
        h2d = sum(d[1] for d in dev_ms) / len(dev_ms)
        d2h = sum(d[2] for d in dev_ms) / len(dev_ms)
        kern_ms = tot - h2d - d2h
        if world == 1:
            workload = ("lcpcomp(coder=huff,threshold=%d,flatten=1,comp=arrays) on %d B %s text (SURVEY 8d generator, seed %d): "
                        "pinned host text -> H2D -> kernels + host Huffman table -> D2H -> stream in pinned host memory, all timed"
                        % (args.threshold, N, args.gen, seed0))
        else:
            workload = ("BASELINE configs[4]: %d independent %d B %s shards (seeds %d+rank), per-shard lcpcomp(coder=huff,threshold=%d,flatten=1); "
                        "%s, all timed"
                        % (world, N, args.gen, seed0, args.threshold,
                           "pinned host text -> H2D -> kernels -> all-gather of the stream sizes -> every rank's D2H to its offset of the block container in shared host memory"
                           if shared is not None else "pinned host text -> H2D -> kernels -> gather of the streams to rank 0 (grouped point-to-point) -> D2H of the block container"))
        line = {
            "metric": "input MB/s end-to-end lcpcomp+huffman",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": N,
                       "parallelism": ("independent shards x%d, %s" % (world, exchange)) if world > 1 else "single GPU"},
            "roofline": roof,
            "hbm_resident": {"value": round(N / 1e6 / (kern_ms * 1e-3), 2), "unit": "MB/s", "ms_per_step": round(kern_ms, 3),
                             "note": "same steps, device time without the H2D / D2H transfers (per GPU)"},
            "pipeline": {"B_alg_bytes_per_input_byte": round(b_alg, 2), "achieved_GBs": round(b_alg * N / (kern_ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(b_alg * N / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "stages_ms": {k2[3:]: round(v, 2) for k2, v in st.items() if k2.startswith("ms_")},
            "stats": {k2: st[k2] for k2 in STAT_KEYS},
            "kernels": {name: {"ms_per_step": round(p["ms"], 3), "launches_per_step": p["launches"],
                               "algorithmic_GBs": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1) if p["ms"] > 0 else None}
                        for name, p in prof.items() if p["launches"]},
        }
        if world == 1 and args.gen == "english" and args.threshold == 2 and N == 2_000_000_000:
            # byte for byte the oracle's stream of the metric's text: size + SHA-256 against the committed full-size golden
            line["stream_matches_golden"] = matches_golden("english_2e9", h_out.a[:out_len])
        if world > 1:
            line["ranks_seen"] = ranks_seen
            line["world_size"] = dist.get_world_size()
            line["gathered_bytes"] = sum(sizes)
            line["exchange"] = exchange
            line["numa_node_rank0"] = numa_node
            line["collective_backend"] = backend
            if container_ok is not None:
                line["container_ok"] = bool(container_ok)
            if rccl is not None:
                line["rccl_gather"] = rccl
        if cpu_res is not None:
            m1, want_len, want_sha = cpu_ref
            sample = np.concatenate([h_text.a[:m1], np.zeros(1, dtype=np.uint8)])
            gpu_prefix, _ = ctx.lcpcomp_compress(sample, args.threshold, 1)
            cpu_res["bit_exact_vs_gpu_on_sample"] = bool(len(gpu_prefix) == want_len and hashlib.sha256(gpu_prefix).hexdigest() == want_sha)
            line["cpu_baseline"] = cpu_res
        if world == 1 and not args.no_extra:
            # the stream of the timed steps back to the text (tdc_gpu_lcpcomp_decompress_into, stream and text in pinned host
            # memory: H2D of the stream, device parse, references, D2H of the text); the result is compared with the input text
            try:
                h_stream = T.PinnedBuffer(out_len)
                h_stream.a[:] = h_out.a[:out_len]
                h_back = T.PinnedBuffer(n)
                ts, nb, dst = [], 0, None
                for i in range(4):
                    t1 = time.perf_counter()
                    nb, dst = ctx.lcpcomp_decompress_into(h_stream, h_back)
                    ts.append(time.perf_counter() - t1)
                t = min(ts[1:])
                line["decompress"] = {"value": round(N / 1e6 / t, 2), "unit": "MB/s of text", "ms": round(t * 1e3, 3), "device_parse": dst["device_parse"],
                                      "rounds": dst["rounds"], "round_trip": bool(nb == n and np.array_equal(h_back.a[:n], h_text.a[:n])),
                                      "note": "the stream of the timed steps through tdc_gpu_lcpcomp_decompress_into, best of 3 after 1 warm-up"}
                h_stream.free(); h_back.free()
            except Exception as e:                       # (an extra never costs the line: e.g. no page-locked memory left for its two buffers)
                line["decompress"] = {"error": str(e)}
        if world == 1 and not args.no_extra and N > (1 << 28):
            m = 1 << 28
            h_text.a[m] = 0                               # the first 256 MiB of the generator's output ARE its 256 MiB text
            ts = []
            for i in range(4):
                t1 = time.perf_counter()
                ol2, st2 = ctx.lcpcomp_compress_into(h_text, m + 1, h_out, args.threshold, 1)
                ts.append(time.perf_counter() - t1)
            t = sum(ts[1:]) / 3
            line["configs1_256MiB"] = {"value": round(m / 1e6 / t, 2), "unit": "MB/s", "ms_per_step": round(t * 1e3, 3),
                                       "device_only_ms": round(st2["ms_total"] - st2["ms_h2d"] - st2["ms_d2h"], 3), "out_len": ol2,
                                       "note": "BASELINE configs[1] through the same end-to-end entry point, 3 steps after 1 warm-up"}
            if args.gen == "english" and args.threshold == 2:
                line["configs1_256MiB"]["stream_matches_golden"] = matches_golden("english_256MiB", h_out.a[:ol2])
        if world == 1 and not args.no_extra and args.gen == "english" and N > (1 << 30):
            # BASELINE.json configs[2]: 10^9 B DNA (sigma = 4), LCPCompressor + ArithmeticCoder, threshold 5 (the compressor's default)
            md = 1_000_000_000
            T.gen_dna(md, 7, out=h_text.a)
            h_text.a[md] = 0
            ts, st3, ol3 = [], None, 0
            for i in range(3):
                t1 = time.perf_counter()
                ol3, st3 = ctx.lcpcomp_compress_into(h_text, md + 1, h_out, 5, 1, T.CODER_ARITH)
                ts.append(time.perf_counter() - t1)
            t = sum(ts[1:]) / 2
            dna = {"value": round(md / 1e6 / t, 2), "unit": "MB/s", "ms_per_step": round(t * 1e3, 3), "out_len": ol3,
                   "stages_ms": {k2[3:]: round(v, 2) for k2, v in st3.items() if k2.startswith("ms_")},
                   "stats": {k2: st3[k2] for k2 in ("factors", "maxlcp", "levels", "small_levels", "purges", "sa_key_words", "sa_mode")},
                   "stream_matches_golden": matches_golden("dna_1e9_arith", h_out.a[:ol3]),
                   "note": "BASELINE configs[2] (SURVEY 8d DNA generator, seed 7) through the same end-to-end entry point, 2 steps after 1 warm-up"}
            if dna_ref is not None:
                m3, want_len3, want_sha3, cpu_s = dna_ref
                sample = np.concatenate([h_text.a[:m3], np.zeros(1, dtype=np.uint8)])
                got3, _ = ctx.lcpcomp_compress(sample, 5, 1, T.CODER_ARITH)
                dna["bit_exact_vs_oracle_on_sample"] = bool(len(got3) == want_len3 and hashlib.sha256(got3).hexdigest() == want_sha3)
                dna["sample"] = "first %d bytes of the DNA text; oracle (1 core) %.1f s" % (m3, cpu_s)
            line["configs2_dna_1e9"] = dna
            # BASELINE.json configs[3]: lz78(coder=gamma).  The parse is sequential by nature (one dependent dictionary step per input byte,
            # compressors/LZ78Compressor.hpp:97-121) and runs on the HOST; only the Elias-gamma packing is a GPU kernel: a host-bound
            # figure, measured on a 32 MiB sample (the full 10^9 B take 69 s; its stream equals the oracle's, tests/golden/oracle_fullsize.json)
            ml = 1 << 25
            lz = T.gen_english(ml, 42)
            t1 = time.perf_counter()
            out3, st4 = ctx.lz78_compress(lz)
            t = time.perf_counter() - t1
            line["configs3_lz78"] = {"value": round(ml / 1e6 / t, 2), "unit": "MB/s", "bound": "host parse (sequential trie walk; not a GPU figure)",
                                     "sample": "first %d bytes of the 10^9 B English text, one call: host parse + device gamma pack %.1f ms" % (ml, st4["ms_total"]),
                                     "phrases": st4["factors"], "out_len": len(out3)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                                     # nobody unmaps the container while rank 0 still reads it
    if shared is not None:
        shared.close(T.host_unregister)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
