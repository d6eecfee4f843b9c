#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X bit-parallel block-DP hot path.

Metric (BASELINE.json): DP cell-updates/sec (GCUPS) + pairs/sec on 100 kbp x 100 kbp pairs at 5 %
divergence, full bit-parallel DP, cost only (configs[1], "C2").

A *step* is one pass of the hot path over one batch of synthetic pairs that is already resident in HBM
as ASCII: BitProfile build kernels -> (round 6) transposes into bit planes over groups of 32 pairs -> reset of the
boundary rows -> the bit-sliced DP kernel (csrc/slice_kernel.hpp: every (group, strip) job) -> per-pair sums ->
read the edit distances back.  `--pairs P` sets the batch per GPU (default 8192 = 256 groups x 32 strips = four jobs per
wave slot; P=1 is the literal single-pair C2 case, which runs on the chained strip kernel, is latency bound on ~49
wavefronts and is reported next to the batch number as `single_pair`).  PA_SLICE=0 runs the batch on the strip kernels of
rounds 1-5 (pair_kernel<8>: 130 TCUPS).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; pairs are independent, so ranks shard the batch with no data-path collective
(weak scaling: every rank aligns its own `--pairs` pairs).  torch is used only for
torch.distributed (RCCL) plumbing: barrier, max-over-ranks of the elapsed time, and a checksum.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the pipelined A*PA2 passes of pa_align run on their own streams: ask the HIP runtime for enough hardware queues BEFORE it starts
# (torch initialises it ahead of the library's own load-time hint)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# The kernel's real bound is VALU issue: 256 CUs x 4 SIMDs x 2.4 GHz, one wave64 instruction per 2 clocks at best.
# Measured issue rates per opcode class and for mixed streams: profiles/r01_runs/issue_probe*.log.
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2
VALU_MIXED_CEILING = 256 * 4 / 1.57e-9
# The headline kernel's own instruction mix (profiles/r05_headline_opcodes.json, from the ISA by tools/opcode_table.py: one K = 8 step is
# 90.5 VALU instructions, 60.6 of the fast class and 29.9 of the slow one) priced with the measured clocks per class: the
# ceiling THIS mix can reach lies between the two -- every slow instruction of the step is a carry of the 256-bit add, a funnel shift across
# subwords, or the per-step plumbing (DPP shift, lane-0 feed), none of which has a fast-class form on gfx950.
STEP_FAST, STEP_SLOW = 60.59, 29.94
CLK_FAST, CLK_SLOW = (2.3, 3.0), (4.1, 4.3)
WEIGHTED_CLK = tuple((STEP_FAST * f + STEP_SLOW * sl) / (STEP_FAST + STEP_SLOW) for f, sl in zip(CLK_FAST, CLK_SLOW))  # (2.90, 3.43)
# The bit-sliced kernel's mix (round 6; ISA of slice_kernel<R>): per strip step 4 R VOP2 with two VGPR sources + 4 R v_bitop3 with three
# (all fast class) + about 12 slow-class (4 v_readlane, 4 DPP, predicates from SGPRs) + about 12 other instructions.  Clocks per class at two
# wavefronts per SIMD from tools/bank_probe (profiles/r06_runs/bank_probe.log: 1.02 / 1.19-1.28 / 1.72 ns at 2.4 GHz nominal); the whole row
# mix with conflict-free register banks runs at 1.00-1.04 ns per instruction there -- the measured ceiling of this instruction stream.
SLICE_CLK = {"vop2": (2.35, 2.45), "bitop3": (2.85, 3.07), "slow": (4.1, 4.3)}
SLICE_ROW_MIX_NS = (1.00, 1.04)


def slice_weighted_clk(rows: int):
    n2, n3, slow, other = 4.0 * rows, 4.0 * rows, 12.0, 12.0
    tot = n2 + n3 + slow + other
    return tuple((n2 * SLICE_CLK["vop2"][i] + other * SLICE_CLK["vop2"][i] + n3 * SLICE_CLK["bitop3"][i] + slow * SLICE_CLK["slow"][i]) / tot for i in (0, 1))


def host_quota_cores() -> int:
    """Cores this process may use: the smaller of its affinity mask and its cgroup's CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def spread(xs):
    """median, min, max and relative spread (max - min) / median of a list of repetitions (host-bound legs vary run to run)."""
    xs = sorted(float(x) for x in xs)
    med = xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])
    return {"median": round(med, 3), "min": round(xs[0], 3), "max": round(xs[-1], 3), "reps": len(xs),
            "rel_spread": round((xs[-1] - xs[0]) / med, 4) if med else 0.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=8192, help="independent pairs per GPU per step")
    ap.add_argument("--seq-len", dest="n", type=int, default=100_000, help="sequence length (bp)")
    ap.add_argument("--div", type=float, default=0.05, help="divergence (edit rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the batched alignment-with-traceback leg")
    ap.add_argument("--no-banded", action="store_true", help="skip the banded leg")
    ap.add_argument("--c4-pairs", type=int, default=10_000)
    ap.add_argument("--c4-strong-pairs", type=int, default=100_000,
                    help="pairs of the C4 strong-scaling leg (the 10 000 C4 pairs repeated): the same for every N, enough for >= 50 ms of GPU work per rank at N = 8")
    ap.add_argument("--host-reps", type=int, default=5, help="repetitions of the host-bound legs (median and spread are reported)")
    ap.add_argument("--no-engine", action="store_true", help="skip the single-pair A*PA2 legs (C3, drop-in loop)")
    ap.add_argument("--no-c5", action="store_true", help="skip the 10 Mbp A*PA2 leg")
    ap.add_argument("--no-apa2", action="store_true", help="skip the batched A*PA2 legs (c4_astarpa2_{simple,full}, c3_batch_*[_full])")
    ap.add_argument("--c3-batch", type=int, nargs="*", default=[512, 4096], help="batch sizes of the 100 kbp batched A*PA2 leg")
    ap.add_argument("--no-c4-sharded", action="store_true", help="skip the C4 strong-scaling leg (sharded_align over all ranks)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")

    # N ranks share one node's cores: every rank's host-side work (upload gathering, SH / GCSH tables) gets its share, not all of them
    quota = host_quota_cores()
    if world > 1:
        os.environ.setdefault("PA_HOST_THREADS", str(max(1, quota // world)))
    import torch

    import astar_pairwise_aligner_amd as pa
    from astar_pairwise_aligner_amd.generate import generate_pair

    pa.require_gpu()
    # PA_BENCH_DRY_MULTI=1 (tests only): exercise the N>1 code path on a one-GPU box -- every rank uses GPU 0 and the
    # collectives run over gloo.  The numbers of such a run mean nothing.
    dry_multi = os.environ.get("PA_BENCH_DRY_MULTI") == "1"
    device_index = 0 if dry_multi else local_rank
    torch.cuda.set_device(device_index)
    pa.capi.load().pa_set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry_multi:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        dist = dist_mod

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic input (seed = global pair index + 1), resident in HBM before timing ----
    pairs = [generate_pair(args.n, args.div, seed=rank * args.pairs + i + 1) for i in range(args.pairs)]
    batch = pa.Batch(pairs)
    st = batch.stats()
    shape = batch.shape()

    for _ in range(args.warmup):
        costs, _ = batch.run()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        costs, ms = batch.run()
        kernel_ms.append(ms)
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        chk = torch.tensor([int(costs.astype("int64").sum())], dtype=torch.int64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        checksum = int(chk.item())
    else:
        checksum = int(costs.astype("int64").sum())

    # ---- C4 strong scaling (BASELINE configs[3]: "10 000 x 10 kbp sharded across the GPUs"): every rank aligns its LPT shard with
    #      traceback on its GPU, one tensor gather of costs + one of CIGAR bytes; total work fixed as N grows ----
    c4_sharded = None
    if not args.no_c4_sharded:
        from astar_pairwise_aligner_amd.sharding import sharded_align

        divs = (0.01, 0.05, 0.10, 0.15)

        # Every rank needs every pair (the queue decides at run time who aligns which chunk), but nobody needs to GENERATE them all: rank r
        # makes the pairs i = r (mod N) and the ranks swap them once (untimed set-up; round 4 generated all 10 000 on every rank).
        own = {i: generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(rank, args.c4_pairs, world)}
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, own)
            for part in parts:
                own.update(part)
            del parts
        c4s = [own[i] for i in range(args.c4_pairs)]
        del own
        from astar_pairwise_aligner_amd.sharding import default_align

        busy = [0.0]

        def timed_align(sub):  # (how long this rank's GPU was given work: the queue's balance, next to the wall time)
            tb = time.perf_counter()
            r = default_align(sub)
            busy[0] += time.perf_counter() - tb
            return r

        from astar_pairwise_aligner_amd import sharding as _sh

        # STRONG scaling: the 10 000 C4 pairs repeated up to --c4-strong-pairs -- the same work for every N, enough that a rank of eight still
        # has tens of milliseconds of GPU work behind the fixed costs (batch creation, one store round trip per chunk, the gather)
        reps_of_c4 = max(1, (args.c4_strong_pairs + args.c4_pairs - 1) // args.c4_pairs)
        strong = c4s * reps_of_c4
        n_strong = len(strong)
        sharded_align(c4s[: 64 * world], all_ranks=False)  # warm-up (buffers, pinned staging)
        times, busies, tims = [], [], []
        res = None
        for rep in range(max(1, args.host_reps if world == 1 else 3)):
            busy[0] = 0.0
            barrier()
            t0s = time.perf_counter()
            res = sharded_align(strong, compute=timed_align, all_ranks=False)  # gathered to rank 0 only: the other ranks return None
            barrier()
            dts = time.perf_counter() - t0s
            tim = dict(_sh.sharded_last_timing)
            busy_all, tim_all = [busy[0]], [tim]
            if dist is not None:
                tt = torch.tensor([dts], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dts = float(tt.item())
                bt_ = torch.zeros(world, dtype=torch.float64, device="cuda")
                bt_[rank] = busy[0]
                dist.all_reduce(bt_, op=dist.ReduceOp.SUM)
                busy_all = [float(x) for x in bt_.tolist()]
                tim_all = [None] * world
                dist.all_gather_object(tim_all, tim)
            times.append(dts)
            busies.append(busy_all)
            tims.append(tim_all)
        if rank == 0:
            best = min(range(len(times)), key=lambda k: times[k])
            per_rank = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in (t or {}).items()} for t in tims[best]]
            c4_sharded = {
                "workload": f"C4 strong scaling: {n_strong} x 10 kbp pairs (the {args.c4_pairs} C4 pairs at 1/5/10/15 %, {reps_of_c4} times), global alignment "
                            f"with traceback, over {world} GPU(s), chunks pulled from a queue (one counter in the process group's store), (cost, CIGAR) of "
                            "every pair gathered to rank 0; the same work for every N",
                "pairs": n_strong,
                "pairs_per_sec": round(n_strong / min(times), 1),
                "ms": round(min(times) * 1e3, 2),
                "ms_repetitions": spread([t * 1e3 for t in times]),
                "n_gpus": world,
                "scaling": "strong",
                "rank_busy_s": [round(x, 4) for x in busies[best]],
                "rank_timing_s": per_rank,
                "host_threads_per_rank": int(os.environ.get("PA_HOST_THREADS", "0")) or quota,
                "cost_checksum": int(sum(c for c, _ in res[: args.c4_pairs])),
                "cigar_bytes": int(sum(len(g) for _, g in res[: args.c4_pairs])),
            }
            assert [c for c, _ in res[: args.c4_pairs]] == [c for c, _ in res[args.c4_pairs: 2 * args.c4_pairs]] or reps_of_c4 < 2
        # WEAK scaling: every rank aligns its OWN 10 000 C4-type pairs (no queue, no exchange but the barrier): N x the work on N GPUs
        wk = [generate_pair(10_000, divs[i % 4], seed=2_000_000 + rank * args.c4_pairs + i) for i in range(args.c4_pairs)]
        default_align(wk[:64])
        wtimes = []
        for rep in range(3):
            barrier()
            t0w = time.perf_counter()
            wres = default_align(wk)
            barrier()
            dtw = time.perf_counter() - t0w
            if dist is not None:
                tt = torch.tensor([dtw], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtw = float(tt.item())
            wtimes.append(dtw)
        wsum = torch.tensor([int(sum(c for c, _ in wres))], dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.all_reduce(wsum, op=dist.ReduceOp.SUM)
        if rank == 0:
            c4_sharded["weak"] = {
                "workload": f"C4 weak scaling: {args.c4_pairs} own 10 kbp pairs per rank (1/5/10/15 %), global alignment with traceback on the rank's GPU, "
                            "nothing exchanged; the slowest rank's time",
                "pairs_per_sec": round(args.c4_pairs * world / min(wtimes), 1), "ms": round(min(wtimes) * 1e3, 2),
                "ms_repetitions": spread([t * 1e3 for t in wtimes]), "n_gpus": world, "scaling": "weak", "cost_checksum": int(wsum.item())}
        del wk, wres
        # the same queue with band-limited work per pair (batched A*PA2 `simple`): what the work of a pair is depends on its divergence
        try:
            from astar_pairwise_aligner_amd.sharding import astarpa2_align

            run_a = astarpa2_align(pa.AstarPa2Params.simple())
            sharded_align(c4s[: 64 * world], compute=run_a, all_ranks=False)
            barrier()
            t0a = time.perf_counter()
            res_a = sharded_align(strong, compute=run_a, all_ranks=False)
            barrier()
            dta = time.perf_counter() - t0a
            if dist is not None:
                tt = torch.tensor([dta], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dta = float(tt.item())
            if rank == 0:
                assert [c for c, _ in res_a] == [c for c, _ in res], "sharded A*PA2 costs differ from the sharded full DP"
                c4_sharded["astarpa2_simple"] = {"pairs": n_strong, "pairs_per_sec": round(n_strong / dta, 1), "ms": round(dta * 1e3, 2),
                                                 "cigar_bytes": int(sum(len(g) for _, g in res_a[: args.c4_pairs]))}
            del res_a
        except AssertionError:
            raise
        except Exception as e:  # (reporting only)
            if rank == 0:
                c4_sharded["astarpa2_simple"] = {"error": str(e)}
        del res, c4s, strong

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    noise = {}  # leg -> relative spread of its repetitions (what the regression check tolerates)
    sliced = int(shape.get("sliced_rows_per_lane", 0))
    wclk = slice_weighted_clk(sliced) if sliced else WEIGHTED_CLK
    total_cells = st["cells"] * world * args.steps
    value = total_cells / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
    achieved_gbs = st["algo_bytes"] / avg_kernel_s / 1e9
    out = {
        "metric": "DP cell-updates/sec (GCUPS), full bit-parallel DP, 100kbp@5% div",
        "value": round(value, 2),
        "unit": "GCUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "pairs_per_sec": round(args.pairs * world * args.steps / elapsed, 2),
        "config": {
            "workload": f"C2: {args.pairs} independent {args.n} bp x {args.n} bp pairs per GPU, {args.div:.0%} divergence, "
                        "full DP cost-only (BitProfile build + " + ("bit-plane transposes + bit-sliced kernel + per-pair sums" if sliced else "strip kernel") +
                        " + cost read-back per step)",
            "pairs_per_gpu": args.pairs,
            "seq_len": args.n,
            "divergence": args.div,
            "strips_per_gpu": int(st["strips"]),
            "word_updates_per_gpu": st["word_updates"],
            "cost_checksum": checksum,
        },
        # What binds the dominant kernel is integer VALU issue (no MFMA, 1.2 GB of algorithmic HBM traffic per 0.6 s launch), so that is
        # what the roofline object prices; the HBM figures the contract defines sit in the same object (hbm_*) and `traffic` is HBM bytes.
        "roofline": {
            "bound": "valu_issue",
            "achieved": round(shape["valu_instructions"] / avg_kernel_s / 1e9, 2),
            "peak": round(VALU_PEAK_WAVE_INSTR / 1e9, 1),
            "unit": "G wave64 VALU instructions/s",
            "frac": round(shape["valu_instructions"] / avg_kernel_s / VALU_PEAK_WAVE_INSTR, 4),
            "traffic": None,
            # the ceiling of the kernel's own opcode mix (see WEIGHTED_CLK above): between the optimistic and the pessimistic class clocks
            "weighted_peak": [round(256 * 4 * 2.4e9 / c / 1e9, 1) for c in wclk],
            "frac_of_weighted_peak": [round(shape["valu_instructions"] / avg_kernel_s / (256 * 4 * 2.4e9 / c), 4) for c in wclk],
            "weighted_clocks_per_instruction": [round(c, 3) for c in wclk],
            "achieved_clocks_per_instruction": round(256 * 4 * 2.4e9 / (shape["valu_instructions"] / avg_kernel_s), 3),
            "kernel": shape["kernel"],
            "kernel_ms_avg": round(avg_kernel_s * 1e3, 4),
            "hbm_bound": "hbm",
            "hbm_achieved": round(achieved_gbs, 4),
            "hbm_peak": HBM_PEAK_GBS,
            "hbm_unit": "GB/s",
            "hbm_frac": round(achieved_gbs / HBM_PEAK_GBS, 8),
            "algorithmic_bytes_per_launch": st["algo_bytes"],
            "note": "achieved = executed wave64 VALU instructions of the launch (ISA model, PMC SQ_INSTS_VALU within 2 %) / HIP-event kernel "
                    "time; peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 clocks per instruction.  hbm_* = SURVEY 8(d): (0.75 B/column + 48 B/word) "
                    "per launch / kernel time against 8 TB/s -- 2e-4 by construction, HBM is not what binds",
        },
        "valu_roofline": {
            "achieved": round(shape["valu_instructions"] / avg_kernel_s / 1e9, 2),
            "peak": round(VALU_PEAK_WAVE_INSTR / 1e9, 1),
            "unit": "G wave64 VALU instructions/s",
            "frac": round(shape["valu_instructions"] / avg_kernel_s / VALU_PEAK_WAVE_INSTR, 4),
            "probe_mixed_stream_rate": round(VALU_MIXED_CEILING / 1e9, 1),
            "ratio_to_probe_mixed_stream": round(shape["valu_instructions"] / avg_kernel_s / VALU_MIXED_CEILING, 4),
            "instructions_per_2048_cells": round(shape["valu_instructions"] / (st["cells"] / 2048.0), 2),
            "note": "(10 + 10k) VALU instructions per 64-lane x 32k-row strip step for k >= 4 (eq words come from LDS: 2 ds_read_b128 per "
                    "k = 8 step, issued a step ahead), (11 + 12k) below (ISA count; PMC SQ_INSTS_VALU is 1.7 % above it). "
                    "peak = 1 instruction / 2 clk / SIMD, reached only by unbroken runs of simple VOP2 / 3-VGPR v_bitop3 ops; "
                    "probe_mixed_stream_rate = what tools/issue_probe measures for a 50/50 stream of those and of carry, "
                    "v_alignbit, v_bfe, DPP or SGPR-operand ops in blocks of >= 8 (1.57 ns per instruction per SIMD); a Myers "
                    "step is 70 % simple ops, so it can sit slightly above that reference stream",
        },
        "batch_shape": {"k": shape["k"], "sequential": shape["sequential"]},
    }
    if sliced:
        out["batch_shape"].update({"sliced_rows_per_lane": sliced, "groups": shape["groups"], "jobs": shape["jobs"]})
        out["roofline"]["model_note"] = (
            "instruction model of the bit-sliced kernel: (8 R + 24) wave64 VALU instructions per strip step (8 per row of 32 pairs x 64 lanes = "
            "2048 cells: 4 VOP2 + 4 v_bitop3).  SURVEY 8(d)'s model (23 u64 ops x 2 issue slots per 64-cell word update) gives a 'fraction' "
            "above 1 on gfx950 for this kernel AND for pair_kernel<8> (v_bitop3 folds the step; here add and shifts are gone altogether), so "
            "it is not used; frac divides executed instructions by the nominal 1 instruction / 2 clk / SIMD at 2.4 GHz")
        out["roofline"]["row_mix_ceiling"] = {
            "ns_per_instruction_per_simd": list(SLICE_ROW_MIX_NS), "g_instr_per_s": [round(1024 / x, 1) for x in SLICE_ROW_MIX_NS],
            "frac_of_it": [round(shape["valu_instructions"] / avg_kernel_s / (1024e9 / x), 4) for x in SLICE_ROW_MIX_NS],
            "note": "tools/bank_probe: the kernel's eight-instruction row, 56 rows long, hand-placed in conflict-free VGPR banks, two wavefronts "
                    "per SIMD -- what this instruction stream can reach on the chip"}
        out["roofline"]["computed_cells_over_nm"] = round(shape["computed_cells"] / st["cells"], 4)
        out["roofline"]["slice_device_bytes"] = shape["device_bytes"]
        out["roofline"]["boundary_bytes_reset_per_step"] = shape["boundary_bytes"]
        out["valu_roofline"]["note"] = ("bit-sliced kernel: (8 R + 24) VALU instructions per strip step of 64 lanes x R rows x 32 pairs (ISA count; "
                                        "see roofline.model_note); probe_mixed_stream_rate is the strip kernels' reference stream")
    out["binding_roofline"] = {k: out["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel")}  # (the name earlier rounds used)
    if c4_sharded is not None:
        out["c4_sharded"] = c4_sharded
        noise["c4_sharded.pairs_per_sec"] = c4_sharded.get("ms_repetitions", {}).get("rel_spread", 0.0)

    # ---- the literal single-pair C2 case (latency bound) ----
    if not args.no_single_pair and args.pairs != 1:
        b1 = pa.Batch(pairs[:1])
        b1.run()
        best = 1e9
        kms = 1e9
        for _ in range(5):
            t = time.perf_counter()
            c1, ms1 = b1.run()
            best = min(best, time.perf_counter() - t)
            kms = min(kms, ms1)
        cells1 = b1.stats()["cells"]
        out["single_pair"] = {"gcups": round(cells1 / best / 1e9, 2), "ms": round(best * 1e3, 4),
                              "kernel_ms": round(kms, 4), "kernel_gcups": round(cells1 / kms / 1e6, 2)}
        b1.close()

    # ---- banded C2: the same pairs inside the diagonal band of the expected divergence (GapGap domain), exact costs ----
    if not args.no_banded and world == 1:
        t = time.perf_counter()
        bb = pa.Batch(pairs, band=args.div + 0.01)
        bcosts, _ = bb.run()
        first = time.perf_counter() - t
        assert bcosts.tolist() == costs.tolist(), "banded costs differ from the full DP"
        best, kms = 1e9, 1e9
        for _ in range(3):
            t = time.perf_counter()
            bcosts, ms1 = bb.run()
            best = min(best, time.perf_counter() - t)
            kms = min(kms, ms1)
        out["banded"] = {
            "workload": f"the same {args.pairs} pairs, cost only, diagonal band for divergence hint {args.div + 0.01:.2f} "
                        "(re-run wider where too narrow; costs identical to the full DP)",
            "pairs_per_sec": round(args.pairs / best, 1),
            "ms": round(best * 1e3, 3),
            "kernel_ms": round(kms, 3),
            "gcups_equivalent": round(bb.stats()["cells"] / best / 1e9, 1),
            "create_plus_first_pass_ms": round(first * 1e3, 1),
            "kernel": bb.shape()["kernel"],
        }
        bb.close()

    # ---- C4 (BASELINE configs[3]): 10 000 x 10 kbp, 1-15 % mixed divergence, cost AND CIGAR on the GPU ----
    if not args.no_c4 and world == 1:
        divs = (0.01, 0.05, 0.10, 0.15)
        c4 = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(args.c4_pairs)]
        bt = pa.Batch(c4, trace=True)
        bt.align()
        best = (1e9, 0.0, 0.0, 0.0)
        for _ in range(3):
            t = time.perf_counter()
            c4_costs, c4_cigars, fwd_ms, tr_ms = bt.align()
            dt = time.perf_counter() - t
            if dt < best[0]:
                best = (dt, fwd_ms, tr_ms, bt.last_c_abi_ms)
        # (parity of this path: tests/test_gpu_batch_align.py; here only the plumbing check that every pair got a CIGAR)
        assert all(len(g) > 0 for g in c4_cigars)
        out["c4_batch_align"] = {
            "workload": f"C4: {args.c4_pairs} independent 10 kbp pairs, 1/5/10/15 % divergence, global alignment with traceback "
                        "(checkpointing forward pass + device-side traceback + CIGAR text), strings delivered to the host",
            "pairs_per_sec": round(args.c4_pairs / best[0], 1),
            "ms": round(best[0] * 1e3, 3),
            "c_abi_ms": round(best[3], 3),  # the pa_batch_align call alone (malloc'ed C strings), before Python decodes them
            "c_abi_pairs_per_sec": round(args.c4_pairs / (best[3] * 1e-3), 1),
            "forward_kernel_ms": round(best[1], 3),
            "trace_kernel_ms": round(best[2], 3),
            "gcups_equivalent": round(bt.stats()["cells"] / best[0] / 1e9, 1),
            "host_engine_fallbacks": bt.trace_fallbacks(),
        }
        bt.close()
        # the same batch with the `simple` preset's traceback options (DT-trace through every block, re-fill where it gives up)
        tp = pa.AstarPa2Params.simple()
        tp.domain, tp.doubling = "full", "none"
        try:
            bd = pa.Batch(c4, trace=True, trace_params=tp)
            bd.align()
            out["c4_batch_align"]["dt_trace_kernel_ms"] = round(min(bd.align()[3] for _ in range(3)), 3)
            bd.close()
        except Exception as e:  # (reporting only)
            out["c4_batch_align"]["dt_trace_kernel_ms"] = f"failed: {e}"

    # ---- PCIe-inclusive rate of the headline workload: host buffers -> device layout -> one pass (never `value`) ----
    if world == 1:
        t = time.perf_counter()
        bp = pa.Batch(pairs)
        bp.run()
        dtp = time.perf_counter() - t
        bp.close()
        out["pcie_inclusive_gcups"] = round(st["cells"] / dtp / 1e9, 1)

    # ---- C3 (BASELINE configs[2]): ONE 100 kbp pair, A*PA2 with traceback through pa_align (the drop-in path).  `simple` runs as
    #      one persistent launch per band-doubling pass with the band logic in the kernel (csrc/sweep_wave.hpp); `full` (GCSH +
    #      pruning + incremental doubling: host-side heuristic) runs one launch per 256-column block ----
    if not args.no_engine and world == 1:
        import oracle as _orc

        a3, b3 = generate_pair(100_000, 0.05, seed=1)
        leg = {}
        for name, mk, oprm in (("simple", pa.AstarPa2Params.simple, _orc.params_simple()), ("full", pa.AstarPa2Params.full, _orc.params_full())):
            al = mk().make_aligner(True)
            al.align(a3, b3)
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                c3, g3, s3 = al.align_with_stats(a3, b3)
                best = min(best, time.perf_counter() - t)
            leg[name] = {"ms": round(best * 1e3, 2), "cost": int(c3), "f_max_tries": int(s3["f_max_tries"]), "blocks": int(s3["num_blocks"]),
                         "computed_lanes": int(s3["computed_lanes"])}
            if not args.no_cpu_baseline:
                t = time.perf_counter()
                wc, wg, _ = _orc.cpu_align(a3, b3, oprm)
                leg[name]["cpu_engine_1core_ms"] = round((time.perf_counter() - t) * 1e3, 2)
                assert (c3, g3) == (wc, wg), "C3: GPU engine and CPU-kernel engine disagree"
        out["c3_engine"] = {"workload": "C3: one 100 kbp x 100 kbp pair, 5 %, A*PA2 with traceback via pa_align (cost + CIGAR), best of 3", **leg}
        # a loop over the astarpa2_simple symbol on C4-shaped pairs: what a relinked astarpa-c user gets
        divs = (0.01, 0.05, 0.10, 0.15)
        loop_pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(200)]
        pa.c_abi_align("astarpa2_simple", *loop_pairs[0])
        loop_rates = []
        for _rep in range(max(1, args.host_reps)):  # host-bound: repeated, the median is the figure and the spread is printed next to it
            t = time.perf_counter()
            got = [pa.c_abi_align("astarpa2_simple", a, b) for a, b in loop_pairs]
            loop_rates.append(len(loop_pairs) / (time.perf_counter() - t))
        sp_loop = spread(loop_rates)
        noise["dropin_loop.pairs_per_sec"] = sp_loop["rel_spread"]
        out["dropin_loop"] = {"workload": "200 x 10 kbp pairs (1/5/10/15 %), one astarpa2_simple() call after the other (cost + CIGAR each); "
                                          f"median of {sp_loop['reps']} repetitions",
                              "pairs_per_sec": round(sp_loop["median"], 1), "pairs_per_sec_repetitions": sp_loop}
        # the same loop from 8 host threads at once (the C ABI is re-entrant; ctypes releases the GIL): what a multi-threaded
        # caller of the drop-in symbol gets from ONE GPU.  Reporting only: never allowed to break the bench line.
        try:
            import threading
            from concurrent.futures import ThreadPoolExecutor

            def threads_rate(T, work_pairs, want):
                gate = threading.Barrier(T)

                def _warm(tid):  # every worker exactly once: device buffers are pooled per host thread
                    gate.wait()
                    pa.c_abi_align("astarpa2_simple", *work_pairs[tid % len(work_pairs)])

                def _work(tid):
                    return [(i, pa.c_abi_align("astarpa2_simple", *work_pairs[i])) for i in range(tid, len(work_pairs), T)]

                with ThreadPoolExecutor(T) as ex:
                    list(ex.map(_warm, range(T)))
                    t = time.perf_counter()
                    parts = list(ex.map(_work, range(T)))
                    dtt = time.perf_counter() - t
                got_t = [r for _, r in sorted(x for part in parts for x in part)]
                assert got_t == want, f"drop-in loop: results under {T} threads differ from the sequential loop"
                return round(len(work_pairs) / dtt, 1)

            # round 5: callers that are inside the symbol at the same time are combined into one batch on the GPU (csrc/engine_hip.hip
            # combine_align); every result is compared with the sequential loop's
            r8 = spread([threads_rate(8, loop_pairs, got) for _ in range(max(1, args.host_reps))])  # (below the combiner's threshold of a dozen concurrent callers)
            out["dropin_loop"]["threads8_pairs_per_sec"] = round(r8["median"], 1)
            out["dropin_loop"]["threads8_repetitions"] = r8
            noise["dropin_loop.threads8_pairs_per_sec"] = r8["rel_spread"]
            many = loop_pairs * 8  # (1600 calls: 25 per thread)
            r64 = spread([threads_rate(64, many, got * 8) for _ in range(max(1, min(3, args.host_reps)))])
            out["dropin_loop"]["threads64_pairs_per_sec"] = round(r64["median"], 1)
            out["dropin_loop"]["threads64_repetitions"] = r64
            noise["dropin_loop.threads64_pairs_per_sec"] = r64["rel_spread"]
            # the same from plain C (tests/c_abi/dropin_threads.c: pthreads, no interpreter lock between the calls; fresh threads per count,
            # every result compared with the single-call route inside the program); reporting only
            try:
                import re as _re
                import shutil
                import subprocess
                import tempfile

                gcc = shutil.which("gcc") or shutil.which("cc")
                libdir = str(ROOT / "astar-pairwise-aligner_amd")
                exe = os.path.join(tempfile.gettempdir(), f"pa_dropin_threads_{os.getpid()}")
                subprocess.run([gcc, "-O2", str(ROOT / "tests" / "c_abi" / "dropin_threads.c"), "-I", str(ROOT / "include"), "-L", libdir, "-lastarpa_c_hip",
                                "-lpthread", "-o", exe], check=True, capture_output=True, timeout=120)
                envc = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
                rc = subprocess.run([exe, "640", "simple", "8", "32", "64"], capture_output=True, text=True, timeout=240, env=envc)
                rates = {int(m.group(1)): float(m.group(2)) for m in _re.finditer(r"threads\s+(\d+):\s+([\d.]+) pairs/s", rc.stdout)}
                out["dropin_loop"]["c_pthreads_pairs_per_sec"] = {str(k): v for k, v in sorted(rates.items())}
                out["dropin_loop"]["c_pthreads_results_equal_single_call"] = rc.returncode == 0
                os.unlink(exe)
            except Exception as e:  # (reporting only)
                out["dropin_loop"]["c_pthreads_pairs_per_sec"] = f"failed: {e}"
            import ctypes as _C

            cc, cb = _C.c_uint64(0), _C.c_uint64(0)
            pa.capi.load().pa_combine_stats(_C.byref(cc), _C.byref(cb))
            out["dropin_loop"]["combined_calls"] = int(cc.value)
            out["dropin_loop"]["combined_batches"] = int(cb.value)
        except AssertionError:
            raise
        except Exception as e:  # (reporting only)
            out["dropin_loop"]["threads8_pairs_per_sec"] = f"failed: {e}"
        if not args.no_cpu_baseline:
            t = time.perf_counter()
            want = [_orc.cpu_align(a, b, _orc.params_simple())[:2] for a, b in loop_pairs[:50]]
            out["dropin_loop"]["cpu_engine_1core_pairs_per_sec"] = round(50 / (time.perf_counter() - t), 1)
            assert got[:50] == want, "drop-in loop: GPU and CPU-kernel engine disagree"

    # ---- C5 (BASELINE configs[4]): ONE 10 Mbp x 10 Mbp pair, 5 %, doubling-band A*PA2 (`simple`), traceback off, through pa_align ----
    if not args.no_c5 and world == 1:
        a5, b5 = generate_pair(10_000_000, 0.05, seed=1)
        al5 = pa.AstarPa2Params.simple().make_aligner(False)
        t = time.perf_counter()
        c5, _, s5 = al5.align_with_stats(a5, b5)
        dt5 = time.perf_counter() - t
        out["c5"] = {"workload": "C5: one 10 Mbp x 10 Mbp pair, 5 %, A*PA2 simple (GapCost band doubling from h0 + 256), traceback off, via pa_align: "
                                 "one persistent launch per pass, j_range / fixed_j_range decided in the kernel",
                     "seconds": round(dt5, 3), "cost": int(c5), "f_max_tries": int(s5["f_max_tries"]), "blocks": int(s5["num_blocks"]),
                     "computed_lanes": int(s5["computed_lanes"]),
                     "gcups_computed": round(s5["computed_lanes"] * 64 * 256 / dt5 / 1e9, 1),
                     "gcups_equivalent": round(len(a5) * len(b5) / dt5 / 1e9, 1),
                     "sanity_violations": int(s5["sanity_violations"])}
        del a5, b5

    # ---- batched A*PA2 (pa_batch_create_params): AstarPa2Params::simple() for many pairs, ONE wavefront per pair runs the whole band
    #      search, the device traceback walks the banded blocks.  Cost, CIGAR and statistics are what a loop over pa_align returns
    #      (tests/test_gpu_apa2_batch.py); here the rates, and a sample checked against the CPU-kernel engine ----
    if not args.no_apa2 and world == 1:
        import oracle as _orc

        divs = (0.01, 0.05, 0.10, 0.15)

        def apa2_leg(ps, what, sample, preset="simple"):
            mk = pa.AstarPa2Params.full if preset == "full" else pa.AstarPa2Params.simple
            oprm = _orc.params_full() if preset == "full" else _orc.params_simple()
            t = time.perf_counter()
            ba = pa.Batch(ps, params=mk())
            t_create = time.perf_counter() - t
            ba.align()
            best = (1e9, 0.0, 0.0, 0.0)
            for _ in range(3):
                t = time.perf_counter()
                cs, gs, f_ms, t_ms = ba.align()
                dt = time.perf_counter() - t
                if dt < best[0]:
                    best = (dt, f_ms, t_ms, ba.last_c_abi_ms)
            t_strings = (1e9, 1e9)
            for _ in range(2):  # the reference-style entry point: one malloc'ed C string per pair (pa_batch_align)
                t = time.perf_counter()
                cs2, gs2, _, _ = ba.align_c_strings()
                dt = time.perf_counter() - t
                if ba.last_c_abi_ms < t_strings[1]:
                    t_strings = (dt, ba.last_c_abi_ms)
                assert list(cs2) == list(cs) and gs2 == gs
            sts = ba.pair_stats()
            for i in sample:  # plumbing check on a sample (the parity tests compare every pair)
                wc, wg, ws = _orc.cpu_align(*ps[i], oprm)
                assert (int(cs[i]), gs[i]) == (wc, wg) and sts[i]["computed_lanes"] == ws["computed_lanes"], f"batched A*PA2 ({preset}) differs from the CPU-kernel engine on pair {i}"
            lanes = float(sum(x["computed_lanes"] for x in sts))
            leg = {
                "workload": what,
                "pairs_per_sec": round(len(ps) / best[0], 1),
                "ms": round(best[0] * 1e3, 3),
                "c_abi_ms": round(best[3], 3),                 # pa_batch_align_view: texts left in the plan's host buffer (what the Python layer calls)
                "c_abi_pairs_per_sec": round(len(ps) / (best[3] * 1e-3), 1),
                "c_abi_strings_ms": round(t_strings[1], 3),     # pa_batch_align: one malloc'ed NUL-terminated string per pair (best of 2)
                "c_abi_strings_pairs_per_sec": round(len(ps) / (t_strings[1] * 1e-3), 1),
                "forward_kernel_ms": round(best[1], 3),
                "trace_kernel_ms": round(best[2], 3),
                "create_ms": round(t_create * 1e3, 1),
                "computed_lanes": lanes,                      # 64-row words x 256-column blocks actually computed (BlockStats)
                "band_fraction_of_matrix": round(lanes * 64 * 256 / ba.stats()["cells"], 4),
                "band_gcups_forward": round(lanes * 64 * 256 / (best[1] * 1e-3) / 1e9, 1),
                "mean_f_max_tries": round(sum(x["f_max_tries"] for x in sts) / len(sts), 2),
                "host_engine_fallbacks": ba.trace_fallbacks(),
                "kernel": ("pa::apa2::apa2_full_kernel" if preset == "full" else "pa::apa2::apa2_kernel") + " + pa::trace_kernel<true, true>",
                # round 5: the pairs start most-expensive-first by a k-mer sketch (csrc/sketch_unit.hip, inside create_ms), and two blocks of
                # at most 16 words from two pairs run as ONE strip when the batch fills the chip (csrc/strip2_kernel.hpp); counted per block
                "start_order": "divergence sketch",
                "half_wave_blocks": ba.rdv_stats(),
            }
            if preset == "full":
                fi = ba.full_info()
                # the matches of GCSH (seeds, k-mer matches, local pruning) are part of the BATCH: found once when it is created -- by the GPU
                # (csrc/gcsh_build_kernel.hpp), or by host threads with PA_GCSH_HOST_BUILD=1 -- i.e. OUTSIDE the timed align(); the contours are
                # derived, probed and pruned on the GPU inside it.  pairs_per_sec is the resident rate, ..._incl_create_again everything.
                if fi["build_ms"] < 0:
                    leg["gpu_match_building_ms"] = round(-fi["build_ms"], 2)
                else:
                    leg["host_match_building_ms"] = round(fi["build_ms"], 1)
                leg["matches"] = fi["matches"]
                leg["pairs_per_sec_align_plus_match_building"] = round(len(ps) / (best[0] + abs(fi["build_ms"]) * 1e-3), 1)
            ba.close()
            creates, alls = [], []
            for _ in range(max(1, min(3, args.host_reps))):  # creation is host work (upload gathering, tables, hipMalloc from the cache): repeated
                t = time.perf_counter()
                bb2 = pa.Batch(ps, params=mk())  # the large device buffers come back from the library's cache
                creates.append((time.perf_counter() - t) * 1e3)
                t = time.perf_counter()
                bb2.align()
                alls.append(creates[-1] * 1e-3 + (time.perf_counter() - t))
                bb2.close()
            sp_c = spread(creates)
            leg["create_again_ms"] = round(sp_c["median"], 1)
            leg["create_again_repetitions"] = sp_c
            leg["pairs_per_sec_incl_create_again"] = round(len(ps) / sorted(alls)[len(alls) // 2], 1)
            return leg

        c4a = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(args.c4_pairs)]
        out["c4_astarpa2_simple"] = apa2_leg(c4a, f"C4: {args.c4_pairs} independent 10 kbp pairs, 1/5/10/15 % divergence, A*PA2 `simple` (band doubling, GapCost, DT-trace): "
                                             "cost + CIGAR + statistics of every pair, strings delivered to the host", range(0, min(args.c4_pairs, 40), 1))
        out["c4_astarpa2_full"] = apa2_leg(c4a, f"C4: {args.c4_pairs} independent 10 kbp pairs, 1/5/10/15 % divergence, A*PA2 `full` (GCSH k = 12 with local pruning, pruning of "
                                           "matches, incremental doubling, DT-trace) in the batch kernels: cost + CIGAR + statistics of every pair", range(0, min(args.c4_pairs, 40), 1), "full")
        del c4a
        # one traced 100 kbp pair through the batch API (goes through the single-pair engine: pa_bitpacking_hip.h)
        one = [generate_pair(100_000, 0.05, seed=3_000_000)]
        b1 = pa.Batch(one, params=pa.AstarPa2Params.simple())
        b1.align()
        t1 = []
        for _ in range(3):
            t = time.perf_counter()
            c1, g1, _, _ = b1.align()
            t1.append(time.perf_counter() - t)
        w1 = _orc.cpu_align(*one[0], _orc.params_simple())
        assert (int(c1[0]), g1[0]) == w1[:2]
        b1.close()
        out["c3_batch_1"] = {"workload": "one 100 kbp pair, 5 %, A*PA2 `simple` with traceback through pa_batch_create_params / pa_batch_align",
                             "ms": round(min(t1) * 1e3, 3), "cost": int(c1[0])}
        for n3 in args.c3_batch:
            c3a = [generate_pair(100_000, 0.05, seed=3_000_000 + i) for i in range(n3)]
            out[f"c3_batch_{n3}"] = apa2_leg(c3a, f"C3 batched: {n3} independent 100 kbp pairs, 5 % divergence, A*PA2 `simple` with traceback", range(0, min(n3, 2)))
            out[f"c3_batch_{n3}_full"] = apa2_leg(c3a, f"C3 batched: {n3} independent 100 kbp pairs, 5 % divergence, A*PA2 `full` with traceback", range(0, min(n3, 2)), "full")
            del c3a

    # ---- CPU baseline: the AVX2 port of the reference's SIMD schedule, 1 core, bounded sample ----
    if not args.no_cpu_baseline:
        import oracle

        a0, b0 = pairs[0]
        want0 = oracle.nw_cost(a0, b0, True)  # doubles as the parity check of pair 0
        assert int(costs[0]) == want0, f"GPU cost {int(costs[0])} != oracle {want0}"
        reps, spent = 0, 0.0
        while spent < args.cpu_seconds and reps < 64:
            a_s, b_s = pairs[reps % len(pairs)]
            t = time.perf_counter()
            oracle.nw_cost(a_s, b_s, True)
            spent += time.perf_counter() - t
            reps += 1
        cpu_cells = sum(len(pairs[i % len(pairs)][0]) * len(pairs[i % len(pairs)][1]) for i in range(reps))
        out["cpu_baseline"] = {
            "value": round(cpu_cells / spent / 1e9, 2),
            "unit": "GCUPS",
            "cores": 1,
            "kind": "port",
            "sample": f"{reps} of the same {args.n} bp pairs, full DP cost-only as 256-column operator calls "
                      "(oracle/strip_avx2.c: AVX2 port of simd::compute::<2,(u64,u64),4>, -O3 -march=native, 1 thread; "
                      "the reference is single-threaded)",
        }
        out["speedup_vs_cpu_1core"] = round(value / out["cpu_baseline"]["value"], 1)
        if "single_pair" in out:  # the literal C2 configuration ("single pair") against the same 1-core baseline
            out["single_pair_speedup_vs_cpu_1core"] = round(out["single_pair"]["gcups"] / out["cpu_baseline"]["value"], 1)

        # ---- the same CPU kernels on every core THIS PROCESS MAY USE: independent pairs, threads inside the oracle
        #      library (one atomic work counter).  os.cpu_count() is not that number on a box with a CPU quota or an affinity mask: a
        #      short scaling probe finds the thread count that still pays, and everything is labelled with the speed-up it measured,
        #      nothing extrapolated.  Bounded (about 15 s in all); reporting only. ----
        try:
            t_leg = time.perf_counter()
            try:
                affinity = len(os.sched_getaffinity(0))
            except (AttributeError, OSError):
                affinity = os.cpu_count() or 1
            cgroup = None
            for cg in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
                try:
                    cgroup = open(cg).read().strip()
                    break
                except OSError:
                    pass
            cpu_model = ""
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            divs = (0.01, 0.05, 0.10, 0.15)
            probe_jobs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(384)]
            probe = {}
            tcount = 1
            while tcount <= min(affinity, 256):
                t = time.perf_counter()
                oracle.cpu_many(probe_jobs, oracle.params_simple(), tcount)
                probe[tcount] = time.perf_counter() - t
                if tcount >= 4 and probe[tcount] > 0.9 * probe[tcount // 2]:  # doubling the threads no longer pays
                    break
                tcount *= 2
            best_t = min(probe, key=probe.get)
            eff = probe[1] / probe[best_t]
            # a CPU quota lets short bursts run on more cores than it sustains: the longer samples below are sized by the quota
            quota = None
            try:
                q, per = (cgroup or "").split()[:2]
                quota = float(q) / float(per) if q != "max" else None
            except (ValueError, IndexError):
                pass
            if quota is not None and quota < eff:
                eff = quota
                best_t = max(1, min(best_t, int(quota + 0.999)))
            nb = {"os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "cgroup_cpu_max": cgroup, "cpu": cpu_model, "kind": "port",
                  "threads_used": best_t, "effective_cores": round(eff, 1), "cgroup_quota_cores": quota,
                  "scaling_probe_pairs_per_sec": {str(k): round(len(probe_jobs) / v, 1) for k, v in probe.items()},
                  "note": "effective_cores = min(best speed-up of the scaling probe over one thread, CPU quota of the cgroup); "
                          "the figures below are what THIS process gets from the host it runs on, not a full socket"}
            # (a) full DP, cost only (the headline workload): whole 100 kbp pairs, about 5 s
            njobs = max(best_t, int(4.0 * eff * out["cpu_baseline"]["value"] * 1e9 / (args.n * args.n)))
            jobs = [pairs[i % len(pairs)] for i in range(njobs)]
            t = time.perf_counter()
            got_cpu = oracle.cpu_many(jobs, None, best_t)
            dtc = time.perf_counter() - t
            assert got_cpu[: len(pairs)] == [int(c) for c in costs[: len(got_cpu)]][: len(pairs)], "all-core CPU baseline disagrees with the GPU costs"
            nb["full_dp_gcups"] = round(sum(len(a) * len(b) for a, b in jobs) / dtc / 1e9, 1)
            nb["full_dp_sample"] = f"{len(jobs)} pairs of {args.n} bp over {best_t} threads"
            nb["gpu_over_effective_cores_full_dp"] = round(value / nb["full_dp_gcups"], 1)
            # (b) C4 through A*PA2 `simple` and `full` with traceback (the CPU-kernel engine), about 3 s each
            for preset, prm in (("simple", oracle.params_simple()), ("full", oracle.params_full())):
                rate1 = len(probe_jobs) / probe[1]
                c4j = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(max(best_t * 8, int(3.0 * eff * rate1)))]
                t = time.perf_counter()
                oracle.cpu_many(c4j, prm, best_t)
                dtc = time.perf_counter() - t
                nb[f"c4_astarpa2_{preset}_pairs_per_sec"] = round(len(c4j) / dtc, 1)
                nb[f"c4_{preset}_sample"] = f"{len(c4j)} of the C4 pairs over {best_t} threads (cost + CIGAR each)"
                if f"c4_astarpa2_{preset}" in out:
                    nb[f"gpu_over_effective_cores_c4_astarpa2_{preset}"] = round(out[f"c4_astarpa2_{preset}"]["pairs_per_sec"] / nb[f"c4_astarpa2_{preset}_pairs_per_sec"], 1)
            if "dropin_loop" in out and isinstance(out["dropin_loop"].get("pairs_per_sec"), float):
                nb["dropin_loop_over_effective_cores"] = round(out["dropin_loop"]["pairs_per_sec"] / nb["c4_astarpa2_simple_pairs_per_sec"], 3)
            nb["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
            out["cpu_baseline_nproc"] = nb
        except Exception as e:  # (reporting only)
            out["cpu_baseline_nproc"] = {"error": str(e)}

    # PMC-derived HBM traffic of the dominant kernel (separate rocprofv3 --pmc passes, tools/pmc_run.sh; committed summary in
    # profiles/pmc_latest.json, stamped with the hash of the library sources it was collected with).  Printed only for THIS code.
    pmc = ROOT / "profiles" / "pmc_latest.json"
    if pmc.exists():
        try:
            from astar_pairwise_aligner_amd import _build

            pj = json.loads(pmc.read_text())
            same = pj.get("batch_shape") == out["batch_shape"]
            current = pj.get("kernel_source_hash") == _build.kernel_hash()
            out["roofline"]["traffic"] = round(pj["hbm_bytes_per_word_update"] * st["word_updates"], 1) if same and current else None
            out["roofline"]["traffic_note"] = (
                "FETCH_SIZE+WRITE_SIZE (KiB*1024) per 64-cell word update of the same batch shape from profiles/pmc_latest.json (separate "
                "rocprofv3 --pmc passes over this kernel's current source) x word updates of this launch; dominated by the 8-byte hand-off "
                "granules, each moving a 32-64 B sector" if same and current else
                "profiles/pmc_latest.json was collected for " + ("another batch shape" if not same else "an older strip_kernel.hpp / slice_kernel.hpp") + ": not printed")
            if same and current:
                out["roofline"]["pmc_valu_instructions"] = pj["counters"].get("SQ_INSTS_VALU", {}).get("avg_per_launch")
                if out["roofline"]["pmc_valu_instructions"]:  # the counter, not the model, prices the fraction when it was collected over this code
                    out["roofline"]["frac_model"] = out["roofline"]["frac"]
                    out["roofline"]["achieved"] = round(out["roofline"]["pmc_valu_instructions"] / avg_kernel_s / 1e9, 2)
                    out["roofline"]["frac"] = round(out["roofline"]["pmc_valu_instructions"] / avg_kernel_s / VALU_PEAK_WAVE_INSTR, 4)
                    out["binding_roofline"] = {k: out["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel")}
        except Exception as e:  # a malformed summary must not break the bench line
            out["roofline"]["traffic_note"] = f"pmc summary unreadable: {e}"

    # ---- regressions: every leg against the reference line (profiles/bench_reference.json: the last committed bench line) ----
    try:
        ref_path = ROOT / "profiles" / "bench_reference.json"
        ref = json.loads(ref_path.read_text())
        watch = [("value", True), ("single_pair.ms", False), ("banded.pairs_per_sec", True), ("c4_batch_align.pairs_per_sec", True),
                 ("c4_batch_align.forward_kernel_ms", False), ("c4_batch_align.trace_kernel_ms", False), ("c4_batch_align.dt_trace_kernel_ms", False),
                 # (the kernel times of the C4 A*PA2 legs are spans over four overlapping chunks since round 4: not comparable, not watched)
                 ("c4_astarpa2_simple.pairs_per_sec", True), ("c4_astarpa2_simple.c_abi_pairs_per_sec", True),
                 ("c4_astarpa2_full.pairs_per_sec", True), ("c3_batch_512.pairs_per_sec", True), ("c3_batch_4096.pairs_per_sec", True),
                 ("c3_batch_4096_full.pairs_per_sec", True), ("c3_engine.simple.ms", False), ("c3_engine.full.ms", False), ("c5.seconds", False),
                 ("dropin_loop.pairs_per_sec", True), ("dropin_loop.threads8_pairs_per_sec", True), ("c4_sharded.pairs_per_sec", True),
                 ("pcie_inclusive_gcups", True)]

        def dig(d, path):
            for k in path.split("."):
                d = d[k]
            return float(d)

        regs = []
        for path, higher in watch:
            try:
                was, now = dig(ref, path), dig(out, path)
            except (KeyError, TypeError, ValueError):
                continue
            worse = (was - now) / was if higher else (now - was) / was
            # a leg that was repeated fires only beyond its own run-to-run spread (host-bound legs: 10-30 %); the others beyond 5 %
            # (the end-to-end rates of the batched legs include the host's creation of 10 000 Python strings: 5-8 % from run to run)
            host_bound = path.startswith(("c4_astarpa2_", "c3_batch_", "c4_batch_align.pairs", "pcie_inclusive"))
            if path == "c5.seconds":
                host_bound = True  # (twelve pipelined passes of one pair, a single run: 1.5-1.8 s from run to run, profiles/README.md)
            limit = max((0.15 if path == "c5.seconds" else 0.10) if host_bound else 0.05, 1.25 * noise.get(path, 0.0))
            if worse > limit:
                regs.append({"leg": path, "reference": was, "now": now, "worse_by_pct": round(100 * worse, 1), "limit_pct": round(100 * limit, 1)})
        out["regressions"] = regs
        out["regressions_noise"] = {k: round(v, 4) for k, v in sorted(noise.items())}
        out["regressions_reference"] = ref.get("_from", "profiles/bench_reference.json")
    except Exception as e:  # (reporting only)
        out["regressions"] = f"no reference line: {e}"

    print(json.dumps(out))
    batch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
