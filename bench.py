#!/usr/bin/env python3
"""bench.py — classified reads/s of the classify -> lineage -> consensus hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU via torch.distributed.run)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic 150-bp reads
(1 % substitutions, either strand) against the synthetic 1 Gbp 16S-like database (666 667 x 1 500 bp,
20 000 genus ancestors, 3 % divergence) with its 7-rank synthetic taxonomy and a synthetic RDP stream.
One STEP = one pass of the whole hot path (DUST masking of the reads, seed + ungapped extension, gapped
extension, grouping, -outfmt 6 ordering, per-subject lineage, consensus arg-max) over one batch of 10 M
reads per GPU; reads, RDP assignments, database, seed index and taxonomy are resident in HBM when the
timed region starts, results stay in HBM.
Multi-GPU: reads are sharded (weak scaling, no data-path collective); the database + index are built on
rank 0 and broadcast ONCE over RCCL before the timed region.

Prints one JSON line (rank 0).  `roofline` is for the longest stage of the step (spec v2: the gapped stage,
k_gapped_rows and its sorting passes, or the seed stage -- they are within a per cent of each other; the other
one's object rides along as `roofline.seed_extend` / `roofline.gapped`): algorithmic bytes per launch
 / its HIP-event duration, against 8 TB/s, with the two roofs that really bound it
measured in the same run (random 64-byte lines per second, vector instructions per second per SIMD).
`cpu_baseline` is the oracle's CPU restatement of the same chain ("port") on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def algorithmic_bytes(n_reads, st, read_len):
    """bytes k_seed_extend's algorithm has to move for one launch, each datum once at its
    natural size, counted by the kernel's own counters."""
    words = (read_len + 31) // 32
    per_read = 2 * 8 * words + 8 + 4         # both packed strands, length/offset, per-read hit count
    per_probe = 8                             # one bucket_off pair
    per_posting = 4 + 4 + 4                   # posting + its context record (database bases left / right of the 16-mer)
    per_survivor = 16 + 8 * (words + 1)       # block-table record (subject, bounds) + the database window of the diagonal
    per_hit = 32
    return (n_reads * per_read + st.probes * per_probe + st.postings * per_posting + st.survivors * per_survivor
            + st.hits * per_hit)


def gapped_algorithmic_bytes(st, read_len):
    """bytes the gapped stage has to move per launch: per HSP the 32-byte record in and out, its key and
    region bytes and its entry in the sorted slot list (written and read), the read strand's letters (2 bits each) and the
    database window around the anchor (read length + 2 x 18 + 48 letters)."""
    per_hsp = 32 + 32 + 2 + 2 * 4 + (read_len + 3) // 4 + (read_len + 2 * 18 + 48 + 3) // 4
    return st.hits * per_hsp


def survey_8d(hits_per_read, reads_per_s_per_gpu):
    """SURVEY.md section 8(d)'s canonical per-read figure (megablast geometry: k = 12 query LUT, database stride 17)
    for the WHOLE path, with its nominal P, C, V and the measured E = H.  This build's direct 16-mer index moves
    far fewer bytes, so this is an equivalent-work rate per GPU, not HBM traffic (the survey flags such a figure)."""
    P, Cn, Vn, H = 278, 980, 970, hits_per_read
    a = 40 + 8 * P + 4 * Cn + 16 * Vn + 64 * H + 32 * H + H * (4 + 8 * 5) + 28 + 16
    gbs = a * reads_per_s_per_gpu / 1e9
    return {"bytes_per_read": a, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "basis": "whole step per GPU, nominal C/V, measured E=H"}


def cpu_baseline(cfg, taxdir, sample):
    """The oracle chain (test infrastructure) timed on this box's host cores."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", odir, "liboracle.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)

    class OCfg(C.Structure):
        _fields_ = [("seed", C.c_uint64), ("n_seq", C.c_int64), ("seq_len", C.c_int32), ("n_genus", C.c_int64),
                    ("read_seed", C.c_uint64), ("read_len", C.c_int32)]

    class Res(C.Structure):
        _fields_ = [("gen_s", C.c_double), ("search_s", C.c_double), ("format_s", C.c_double),
                    ("taxcollect_s", C.c_double), ("consensus_s", C.c_double), ("reads", C.c_int64), ("hits", C.c_int64),
                    ("recs", C.c_int64), ("threads", C.c_int32)]
    oc = OCfg(cfg.seed, cfg.n_seq, cfg.seq_len, cfg.n_genus, cfg.read_seed, cfg.read_len)
    threads = max(1, min(os.cpu_count() or 1, 16))
    lib.o_bench_chain.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_char_p, C.c_void_p]
    # two runs on half the sample each, the faster one reported (the figure swung 11.5-15.9 k between runs of round 1)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    best, runs = None, []
    half = max(1, sample // 2)
    for k in range(2):
        res = Res()
        rc = lib.o_bench_chain(C.byref(oc), k * half, half, threads, taxdir.encode(), C.byref(res))
        if rc != 0:
            return None
        t = res.search_s + res.format_s + res.taxcollect_s + res.consensus_s
        runs.append(half / t)
        if best is None or t < best[0]:
            best = (t, res.search_s, res.format_s, res.taxcollect_s, res.consensus_s)
    t, se, fo, tc, co = best
    return {"value": half / t, "unit": "reads/s", "cores": threads, "kind": "port", "runs": runs,
            "sample": "best of 2 runs of %d reads of the same stream vs the full database (spec v2: gapped): search %.1fs + format "
                      "%.1fs + taxcollector %.1fs (%d threads each) + consensus %.1fs (one thread: a cursor walk)" % (
                          half, se, fo, tc, threads, co)}


def inclusive(pg, _capi, cfg, db, tmp, first, n):
    """File to file on a bounded sample: a FASTA file and an RDP file in (page cache), the Consensus text out -- the rate a
    user of the command lines sees.  The resident rate (`value`) leaves out everything measured here except the kernels."""
    fa, rf, out = os.path.join(tmp, "incl_reads.fa"), os.path.join(tmp, "incl_rdp.txt"), os.path.join(tmp, "incl_consensus.txt")
    reads = pg.Reads.from_synth(cfg, first, n)
    rdp = pg.Rdp.from_synth(cfg, first, n, db)
    reads.write_fasta(fa)
    rdp.write_file(rf, reads, db)
    del reads, rdp
    t = [time.perf_counter()]
    r = pg.Reads.from_fasta(fa)
    t.append(time.perf_counter())
    p = pg.Rdp.from_file(rf, r, db)
    t.append(time.perf_counter())
    hits, recs = _capi.classify_consensus(db, r, p)
    t.append(time.perf_counter())
    st = _capi.stage_times()
    n_text = _capi.consensus_format_file(db, r, hits, recs, out)   # rendered on the device, written piece by piece
    t.append(time.perf_counter())
    size = os.path.getsize(fa) + os.path.getsize(rf)
    for x in (fa, rf, out):
        os.remove(x)
    return {"value": n / (t[-1] - t[0]), "unit": "reads/s",
            "sample": "%d reads: %.0f MB of FASTA + RDP text in, %.0f MB of consensus text out" % (n, size / 1e6, n_text / 1e6),
            "stages_s": {"fasta_to_hbm": t[1] - t[0], "rdp_to_hbm": t[2] - t[1], "classify_consensus": t[3] - t[2],
                         "consensus_text_to_file": t[4] - t[3]},
            "classify_consensus_kernels_ms": st.total_ms, "classify_consensus_attempts": st.attempts}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=200000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inclusive-sample", type=int, default=2_000_000, help="reads of the file-to-file line (0: skip it)")
    ap.add_argument("--n-seq", type=int, default=666667)
    ap.add_argument("--ungapped", action="store_true", help="blastn -ungapped: stop after the ungapped stage (spec v1, round 1's tables)")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="set-up only: build / broadcast / import the database, print one line per rank (device, free HBM, broadcast and "
                         "rebuild seconds, database checksums) and fail with the rank's number if a rank's copy differs; no timed steps")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU (nor imported torch), so it
        # may start the ranks itself -- as a CHILD process whose output and exit status it relays (the `mpirun -np N` of
        # Scripts/submit_MPI-blast.job:24 for this build); never an exec after GPU initialisation
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world_env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    # test aid for a one-GPU box: PGX_BENCH_ONE_DEVICE=1 puts every rank on device 0 and PGX_BENCH_BACKEND=gloo replaces
    # RCCL (which refuses two ranks on one device), so that the N > 1 code path can be exercised end to end
    if os.environ.get("PGX_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("PGX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda:%d" % local_rank)  # torch's HIP context first (RCCL needs it)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    pg.init(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = pg.SynthCfg.default()
    if args.n_seq != cfg.n_seq:  # smaller database for quick functional runs
        cfg.n_genus = max(1, cfg.n_genus * args.n_seq // cfg.n_seq)
        cfg.n_seq = args.n_seq
    # one taxonomy directory per job: rank 0 writes the dumps and builds the .bin files (tax_class -c), the others open
    # them after a barrier -- eight ranks each building the same tables was eight times the host work for nothing
    tmp = tempfile.mkdtemp(prefix="pgx_bench_") if rank == 0 else None
    if world > 1:
        box = [tmp]
        dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    try:
        # ---- setup (untimed): taxonomy, database + seed index, one-off broadcast, binding, batches
        if rank == 0:
            _capi._check(pg.lib().pgx_synth_write_taxdump(C.byref(cfg), tmp.encode()))
            pg.TaxDb.create(tmp)
        if world > 1:
            dist.barrier()
        tax = pg.TaxDb.open(tmp)
        t0 = time.time()
        if rank == 0:
            db = pg.Db.from_synth(cfg)
        t_index = time.time() - t0
        t_bcast, bcast_bytes, t_rebuild = 0.0, 0, 0.0
        if world > 1:
            from pangea_plus_amd import sharding
            torch.cuda.synchronize()
            t0 = time.time()
            sent = []

            def arrays_of(d):  # zero-copy aliases of library-owned HBM
                out = [(n, torch.as_tensor(v, device=dev)) for n, v in d.device_arrays()]
                sent.extend(t.numel() for _, t in out)
                return out

            t_fin = [0.0]

            def finish(d):  # the receiving ranks build block tables + seed index from the packed bases they were sent
                torch.cuda.synchronize()
                t1 = time.time()
                d.finish_import()
                t_fin[0] = time.time() - t1
            db = sharding.broadcast_database(db if rank == 0 else None, rank, world, dist, pg.Db.alloc_like, arrays_of, finish)
            torch.cuda.synchronize()
            t_bcast = time.time() - t0 - t_fin[0]
            bcast_bytes = int(sum(sent))
            tt = torch.tensor([t_bcast, t_fin[0]], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_bcast, t_rebuild = float(tt[0].item()), float(tt[1].item())
        if args.dry_ranks:
            # what the first multi-GPU run should be: every rank reports its device and its copy of the database
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            mine = {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank), "hbm_free_GiB": free_b / 2**30,
                    "hbm_total_GiB": total_b / 2**30, "broadcast_s": t_bcast, "index_rebuild_s": t_rebuild, "checksum": list(db.checksum())}
            bad = 0
            if world > 1:
                # the 64-bit sums as three non-negative pieces each (bits 0-21, 22-43, 44-63: the collective is on int64 tensors)
                cs = torch.tensor([(x >> sh) & 0x3FFFFF for x in mine["checksum"] for sh in (0, 22, 44)], dtype=torch.int64, device=dev)
                rank0 = cs.clone()
                dist.broadcast(rank0, src=0)
                bad = int(not torch.equal(cs, rank0))
                rows = [None] * world
                dist.all_gather_object(rows, mine)
                flag = torch.tensor([bad], dtype=torch.int64, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.SUM)
                n_bad = int(flag.item())
            else:
                rows, n_bad = [mine], 0
            if rank == 0:
                for r in rows:
                    print(json.dumps(r), flush=True)
                print(json.dumps({"dry_ranks": world, "ranks_that_differ_from_rank_0": n_bad,
                                  "ranks": [r["rank"] for r in rows if r["checksum"] != rows[0]["checksum"]]}), flush=True)
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            if bad:
                sys.exit("bench.py --dry-ranks: rank %d holds a different database than rank 0" % rank)
            return
        db.bind_taxonomy(tax)
        if args.ungapped:
            db.set_ungapped(True)
        # spec S3d inside every step: the DUST window bits of the batch are computed again by every search, as BLAST does
        db.set_dust_each_search(True)
        from pangea_plus_amd.sharding import batch_first_read as sharding_first
        B = args.reads
        batches = []
        n_batches = min(args.warmup + args.steps, 8)  # a ring of distinct resident batches (1.1 GB each) bounds the memory
        t_import = []
        for s in range(n_batches):
            first = sharding_first(s, rank, world, B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reads = pg.Reads.from_synth(cfg, first, B)
            torch.cuda.synchronize()
            t_import.append(time.perf_counter() - t0)
            rdp = pg.Rdp.from_synth(cfg, first, B, db)
            batches.append((reads, rdp))

        def step(i):
            reads, rdp = batches[i % len(batches)]
            _capi.classify_consensus(db, reads, rdp, want_records=False, want_hits=False)
            return _capi.stage_times()

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # set-up, untimed like the rest of it: size the persistent workspaces and bring the GPU out of the idle clocks
        # the host-side taxonomy binding leaves it in (a few pipeline passes; the W warm-up steps follow)
        for _ in range(3):
            step(0)
        for i in range(args.warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        kernel_ms, alg_bytes, gap_ms, gap_bytes, stages = 0.0, 0, 0.0, 0, []
        for i in range(args.warmup, args.warmup + args.steps):
            st = step(i)
            kernel_ms += st.seed_extend_ms
            alg_bytes += algorithmic_bytes(B, st, cfg.read_len)
            gap_ms += st.gapped_ms
            gap_bytes += gapped_algorithmic_bytes(st, cfg.read_len)
            stages.append(st)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # the same K steps with the DUST bits of the batch import reused (the search then skips its first stage): a side line
        fence()
        db.set_dust_each_search(False)
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            step(i)
        fence()
        dt_nodust = time.perf_counter() - t0
        db.set_dust_each_search(True)
        if world > 1:
            tt = torch.tensor([dt_nodust], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_nodust = float(tt.item())

        if rank == 0:
            last = stages[-1]
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            traffic, gap_traffic = None, None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp))
                    traffic, gap_traffic = tj.get("k_seed_extend_bytes_per_launch"), tj.get("k_gapped_stage_bytes_per_launch")
                    if not (B == 10_000_000 and args.n_seq == pg.SynthCfg.default().n_seq):
                        traffic = gap_traffic = None  # the counter figures are for the default launch only
                except Exception:
                    traffic = gap_traffic = None
            # the roof of the kernel's access shape, measured here and now: 64-byte lines per second for random 8-byte
            # lane loads over a 16 GiB table (pgx_probe_gather), against the kernel's L2 requests per second (requests
            # per read from the PMC passes of profiles/, x reads, / measured kernel time)
            line_roof = None
            try:
                lps, pms = C.c_double(), C.c_double()
                _capi._check(pg.lib().pgx_probe_gather(16 << 30, 0, C.byref(lps), C.byref(pms)))
                per_read = json.load(open(tp)).get("k_seed_extend_l2_requests_per_read")
                if per_read and lps.value > 0:
                    req = per_read * B * args.steps / (kernel_ms * 1e-3)
                    line_roof = {"roof_lines_per_s": lps.value, "roof_as_GBps_of_64B_lines": lps.value * 64 / 1e9,
                                 "kernel_l2_requests_per_s": req, "frac": req / lps.value,
                                 "basis": "random 8-byte lane loads, 16 GiB table, 8 loads in flight per lane; kernel requests = %.1f per read (TCC_HIT+TCC_MISS, profiles/traffic.json)" % per_read}
            except Exception as e:  # an older library without the probe: the line is still printed
                line_roof = {"error": str(e)}
            out = {
                "metric": "classified reads/sec (+ HBM GB/s fraction), 150 bp vs 1 Gbp db, 1/2/4/8 GPU",
                "value": world * B * args.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u64 (2-bit packed bases, integer scores)",
                "data": "synthetic",
                "config": {"workload": ("10M synthetic 150 bp reads per GPU vs 1 Gbp nt-slice, %d GPU, full classify->tax_class->consensus" % world)
                                       if (B == 10_000_000 and cfg.n_seq == 666667) else "reduced functional run",
                           "reads_per_gpu_per_step": B, "db_bases": int(cfg.n_seq) * int(cfg.seq_len), "db_seqs": int(cfg.n_seq),
                           "read_len": int(cfg.read_len), "parallelism": "read-sharded x%d, index broadcast once over RCCL" % world,
                           "spec": "pgx-blastn v1 (-ungapped)" if args.ungapped else "pgx-blastn v2 (gapped)"},
                "dust_at_import_only": {"value": world * B * args.steps / dt_nodust, "unit": "reads/s", "ms_per_step": 1e3 * dt_nodust / args.steps,
                                        "note": "side line: the same steps reusing the DUST bits made when the batch was imported (pgx_db_set_dust_each_search(0)); `value` computes them inside every step"},
                "roofline": None,
                "stages_ms_last_step": {"dust": last.dust_ms, "seed_extend": last.seed_extend_ms, "gapped": last.gapped_ms, "gapped_listed_hsps": last.gapped_wide,
                                        "group": last.group_ms,
                                        "sort_consensus": last.sort_ms, "total": last.total_ms},
                "per_read_last_step": {"probes": last.probes / B, "postings": last.postings / B,
                                       "filter_survivors": last.survivors / B, "seed_runs": last.candidates / B, "hits": last.hits / B},
                "setup_s": {"db_generate_and_index": t_index, "index_broadcast": t_bcast, "index_broadcast_bytes": bcast_bytes,
                            "index_rebuild_on_receivers": t_rebuild if world > 1 else 0.0,
                            "read_batch_import_ms": 1e3 * min(t_import),
                            "read_batch_import_note": "per resident batch, before the timed region: synthetic letters, both strands, the search classes (which reads hold a DUST-masked base); the DUST window bits themselves are computed again inside every timed step",
                            "broadcast_mode": "whole index" if os.environ.get("PGX_BCAST_INDEX", "0") not in ("", "0") else "packed bases + offsets, index rebuilt per GPU"},
            }
            seed_roof = {"bound": "hbm", "kernel": "k_seed_extend", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "alg_bytes_per_launch": alg_bytes / args.steps,
                         "kernel_ms_per_launch": kernel_ms / args.steps,
                         "survey_8d": survey_8d(last.hits / B, world * B * args.steps / dt / world), "random_line_roof": line_roof}
            if gap_ms > 0.0:
                # spec v2: the gapped stage (k_gapped_rows + its counting sorts + the list tiers) is the longest stage of the
                # step, or within a per cent of the seed stage.  It is integer work on letters held in LDS and registers (no MFMA), so its share of the streaming
                # HBM roof is small by nature; the line says so instead of hiding the stage behind the seed stage's.  What
                # bounds it, both measured in this run: (1) random 64-byte lines per second -- every HSP's record, read strand
                # and result are single-use lines (the database words stay in L2 because the HSPs are handled region by
                # region); (2) vector instructions per second per SIMD (pgx_probe_issue: two-operand adds / ands / shifts
                # issue about twice as fast as everything else on this chip).
                g_ach = gap_bytes / (gap_ms * 1e-3) / 1e9
                tj = {}
                try:
                    tj = json.load(open(tp))
                except Exception:
                    pass
                full = B == 10_000_000 and args.n_seq == pg.SynthCfg.default().n_seq
                lines_roof = None
                try:
                    if line_roof and "roof_lines_per_s" in line_roof and tj.get("k_gapped_rows_l2_misses_per_read") and full:
                        miss = tj["k_gapped_rows_l2_misses_per_read"] * B * args.steps / (tj.get("k_gapped_rows_share_of_stage", 1.0) * gap_ms * 1e-3)
                        lines_roof = {"roof_lines_per_s": line_roof["roof_lines_per_s"], "kernel_l2_misses_per_s": miss,
                                      "frac": miss / line_roof["roof_lines_per_s"],
                                      "basis": "TCC_MISS of k_gapped_rows per read (profiles/traffic.json: %.1f) x reads / the kernel's share of the stage time, against pgx_probe_gather" % tj["k_gapped_rows_l2_misses_per_read"]}
                except Exception:
                    lines_roof = None
                issue = None
                try:
                    fast, slow = (C.c_double * 4)(), (C.c_double * 4)()
                    _capi._check(pg.lib().pgx_probe_issue(4, 0, fast))   # v_add_u32: the two-operand kind
                    _capi._check(pg.lib().pgx_probe_issue(4, 1, slow))   # v_max3_i32: every other kind
                    valu_per_read = tj.get("k_gapped_rows_valu_instructions_per_read")
                    issue = {"roof_fast_kind_per_s_per_simd": fast[0], "roof_other_kinds_per_s_per_simd": slow[0]}
                    if valu_per_read and full:
                        per_s_per_simd = valu_per_read * B * args.steps / (tj.get("k_gapped_rows_share_of_stage", 1.0) * gap_ms * 1e-3) / 1024
                        issue.update({"valu_instructions_per_read": valu_per_read, "kernel_valu_per_s_per_simd": per_s_per_simd,
                                      "busy_if_all_fast": per_s_per_simd / fast[0], "busy_if_all_other": per_s_per_simd / slow[0],
                                      "basis": "SQ_INSTS_VALU of k_gapped_rows per read (profiles/) x reads / (1024 SIMDs x kernel time), against the measured issue rates of the two kinds of vector instruction at 4 wavefronts per SIMD"})
                except Exception as e:
                    issue = {"error": str(e)}
                gap_roof = {"bound": "hbm", "kernel": "k_gapped_rows (+ k_reg_* / k_seg_* sorting passes, list tiers)", "achieved": g_ach,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g_ach / HBM_PEAK_GBS, "traffic": gap_traffic,
                            "alg_bytes_per_launch": gap_bytes / args.steps, "kernel_ms_per_launch": gap_ms / args.steps,
                            "note": "the stage's real roofs are the random-line rate and the issue rate, both measured here: see DESIGN.md sections 6-7",
                            "random_line_roof": lines_roof, "issue": issue}
                # the longer of the two stages is the line's `roofline`, the other one rides along (since the end of round 3
                # they are within a per cent of each other and change places from run to run)
                if gap_ms > kernel_ms:
                    out["roofline"] = dict(gap_roof, seed_extend=seed_roof)
                else:
                    out["roofline"] = dict(seed_roof, gapped=gap_roof)
            else:
                out["roofline"] = seed_roof  # (-ungapped: spec v1 has no gapped stage)
            if world == 1 and args.inclusive_sample > 0 and not args.no_cpu_baseline:
                try:
                    # two passes over the same files, the second is the line (as `cpu_baseline` is a best of two): the first one
                    # also pins the process's copy buffers and builds the handle's formatter tables, which a job pays once,
                    # not per batch; it is kept beside the line as `first_pass`
                    cold = inclusive(pg, _capi, cfg, db, tmp, 50_000_000, min(args.inclusive_sample, B))
                    out["inclusive"] = inclusive(pg, _capi, cfg, db, tmp, 50_000_000, min(args.inclusive_sample, B))
                    out["inclusive"]["first_pass"] = {"value": cold["value"], "stages_s": cold["stages_s"]}
                except Exception as e:  # the line is still printed
                    out["inclusive"] = {"error": str(e)}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg, tmp, args.cpu_sample)
            else:
                out["cpu_baseline"] = None
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    finally:
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
