"""Benchmark of the codec-decode hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one chunked decode of the batch `BASELINE.json.configs[1]` names: 64 utterances x 30 s (375 code frames
each; chunks of 300 + 25 frames of recomputed left context, the reference's chunked_decode semantics), bf16 storage
and MFMA operands with fp32 accumulation, synthetic uniform codes already resident in HBM, seeded random-init weights
of the flagship architecture.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank decodes
its own 64 x 30 s batch (weak scaling, no collective on the data path); the value is all ranks' audio-seconds over
the slowest rank's time.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      the dominant kernel (by summed time): algorithmic FLOPs of its launches / their summed duration,
                both measured in this run by HIP events on the launch stream, against the dense bf16 MFMA peak;
  cpu_baseline  the CPU restatement of the decode (oracle/, fp32 torch) timed on this host's cores on a bounded
                sample (N = 1 only);
  kernels       the eight kernels with the most time: launches, ms per step, achieved TFLOP/s or GB/s of algorithmic work;
  config        beside the workload: the rate with the samples copied to the host (`value_incl_d2h`) and through the
                `decode` wrapper with numpy results (`value_decode_wrapper`), the step with every context row computed;
  streaming, mixed_lengths, long_form   (N = 1 only) short records of BASELINE.json configs[2..4], measured after and
                outside the timed region; `--workload stream | mixed | longform | encode` are the full measurements,
                `--workload stream --sweep 256,1024,...` the capacity curve.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_command(argv, env):
    """`--gpus N` with N > 1 and no launcher environment: the command that starts N ranks of this script (one per GPU,
    RCCL rendezvous on 127.0.0.1), or None when this process is already a rank (or N == 1)."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1 or "WORLD_SIZE" in env:
        return None
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={known.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def check_world(gpus: int, world: int) -> None:
    """A run must use exactly the GPUs it was asked for: never report an N-GPU line from another number of ranks."""
    if gpus != world:
        raise SystemExit(f"bench.py: --gpus {gpus} but {world} rank(s) are running (WORLD_SIZE); refusing to report a mislabelled result")


if __name__ == "__main__":
    # Start the ranks BEFORE anything in this process can touch a GPU (torch is not even imported yet): the parent only
    # waits for the children - rank 0 prints the JSON line on the shared stdout - and passes their exit status on.
    _cmd = spawn_command(sys.argv[1:], os.environ)
    if _cmd is not None:
        sys.exit(subprocess.run(_cmd).returncode)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import qwen3_tts_amd as q3  # noqa: E402
from qwen3_tts_amd.decoder import Decoder  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/pmc_traffic.py from rocprofv3 --pmc passes


def pmc_traffic(variant: str):
    """HBM bytes per launch of a contraction-kernel variant from the committed PMC passes (None if not collected)."""
    m = __import__("re").match(r"convgemm<(\w+),(\w+),(resunit )?(\d+)x(\d+),k(\d+)B,(\d+)w>", variant)
    if not m or not os.path.exists(PMC_TRAFFIC):
        return None
    ta, to, fused, bm, bn, kb, nw = m.groups()
    mang = {"f32": "f", "bf16": "DF16b"}
    # (rows, columns, waves, fused) -> the kernel template's wave grid and per-wave tile counts
    shapes = {("256", "96", "4", False): (4, 1, 2, 3), ("256", "128", "8", False): (4, 2, 2, 2), ("128", "128", "4", False): (2, 2, 2, 2),
              ("256", "32", "4", False): (4, 1, 2, 1), ("256", "64", "8", False): (8, 1, 1, 2), ("256", "256", "8", False): (4, 2, 2, 4), ("256", "192", "8", False): (4, 2, 2, 3),
              ("256", "192", "8", True): (4, 2, 2, 3), ("128", "192", "4", True): (2, 2, 2, 3), ("256", "96", "4", True): (4, 1, 2, 3), ("256", "32", "4", True): (4, 1, 2, 1),
              ("768", "96", "12", True): (12, 1, 2, 3)}
    sh = shapes.get((bm, bn, nw, bool(fused)))
    if sh is None:
        return None
    if (bm, bn, nw, kb) == ("256", "128", "8", "64"):
        sh = (4, 2, 2, 2)
    key = f"{mang[ta]}{mang[to]}Li{sh[0]}ELi{sh[1]}ELi{sh[2]}ELi{sh[3]}ELi{kb}ELb{1 if fused else 0}ELb0E"
    rec = json.load(open(PMC_TRAFFIC)).get("kernels", {}).get(key)
    return None if rec is None else rec["hbm_bytes_per_launch"]


def cpu_baseline(cfg, weights, batch: int, frames: int, threads: int) -> dict:
    """The oracle (a port: the reference's own decode is in a package that is not installed) on the host cores,
    on a bounded sample of the same workload: `batch` utterances of `frames` frames through the same chunked decode."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import decoder_oracle as oracle
    import statistics
    torch.set_num_threads(threads)
    # BASELINE.json configs[0] as stated: one 10 s utterance, fp32, batch 1; 3 warm-up + 5 timed runs, median
    c0 = q3.random_codes(cfg, 1, 125, seed=0)
    t0s = []
    for i in range(8):
        t1 = time.perf_counter()
        w0 = oracle.chunked_decode(cfg, weights, c0, cfg.chunk_size, cfg.left_context_size)
        if i >= 3:
            t0s.append(time.perf_counter() - t1)
    dt0 = statistics.median(t0s)
    assert w0.shape[-1] == cfg.chunked_output_length(125)
    # the benchmark's own workload on a bounded sample (the threads are warm by now): one timed run
    codes = q3.random_codes(cfg, batch, frames, seed=0)
    t0 = time.perf_counter()
    wav = oracle.chunked_decode(cfg, weights, codes, cfg.chunk_size, cfg.left_context_size)
    dt = time.perf_counter() - t0
    assert wav.shape[-1] == cfg.chunked_output_length(frames)
    return {"value": round(batch * frames / cfg.frame_rate / dt, 4), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "sample": f"{batch} utterances x {frames / cfg.frame_rate:.1f} s ({frames} frames each) of the benchmark's workload, chunked "
                      f"decode 300+25, fp32 torch CPU restatement of the decode (reference package unavailable offline), 1 timed run after "
                      f"the configs[0] runs ({dt:.1f} s)",
            "config0": {"value": round(10.0 / dt0, 4), "unit": "audio-sec/sec", "seconds": round(dt0, 3),
                        "sample": "BASELINE.json configs[0]: 1 utterance x 10 s (125 frames), fp32, batch 1; 3 warm-up + 5 timed runs, median"}}


def stream_bench(args, cfg, weights, device, dec=None) -> dict:
    """BASELINE.json configs[2]: `--streams` concurrent streams, each advanced `--push-frames` frames (80 ms of audio
    per frame) per step in ONE launch sequence, causal state carried in HBM; streams start at staggered positions.
    Latency of a step = host wall time from submitting the push to its samples being complete on the device."""
    import numpy as np
    S, n = args.streams, args.push_frames
    if dec is None:
        dec = Decoder(cfg, weights, device=device, precision=args.precision, max_positions=max(8192, (args.steps + args.warmup + 16) * n))
    pool = dec.open_streams(S)
    ids = pool.open_many(S)
    ids_dev = torch.tensor(ids, dtype=torch.int32, device=device)
    for j in range(1, 8):                                   # stagger: stream i starts (i % 8) frames ahead
        sub = [i for i in ids if i % 8 >= j]
        if sub:
            pool.push_raw(torch.tensor(sub, dtype=torch.int32, device=device), q3.random_codes(cfg, len(sub), 1, seed=100 + j).to(device))
    codes = [q3.random_codes(cfg, S, n, seed=2 + j).to(device) for j in range(8)]
    out = torch.empty((S, n * cfg.total_upsample), dtype=torch.float32, device=device)
    use_graph = not args.no_graph       # the push replayed from a HIP graph per (ids, codes, out) buffer set: 8 graphs here
    for j in range(max(args.warmup, 8)):
        pool.push_raw(ids_dev, codes[j % 8], out, graph=use_graph)
    torch.cuda.synchronize(device)
    lat = []
    t_all = time.perf_counter()
    for j in range(args.steps):
        t0 = time.perf_counter()
        pool.push_raw(ids_dev, codes[j % 8], out, graph=use_graph)
        torch.cuda.synchronize(device)
        lat.append((time.perf_counter() - t0) * 1e3)
    elapsed = time.perf_counter() - t_all
    dec._check(dec._lib.q3c_check_errors(dec._ctx, dec._stream()))
    assert bool(torch.isfinite(out).all())
    lat = np.asarray(lat)
    audio = S * n / cfg.frame_rate * args.steps
    # one profiled step for the kernel breakdown
    dec.profile(True)
    pool.push_raw(ids_dev, codes[0], out)
    torch.cuda.synchronize(device)
    prof = sorted(dec.profile_read(), key=lambda p: -p["ms"])
    dec.profile(False)
    state_bytes = pool.state_bytes_per_stream
    pool.close()
    return {"metric": "audio-sec decoded/sec", "value": round(audio / elapsed, 1), "unit": "audio-sec/sec", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"streaming decode: {S} concurrent streams x {n} frame(s) ({n * 80} ms of audio) per step, causal "
                                   f"state carried in HBM, staggered positions, flagship architecture, random-init weights",
                       "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                       "max_ms": round(float(lat.max()), 3), "real_time_factor": round(n * 80.0 / float(np.percentile(lat, 99)), 1),
                       "state_bytes_per_stream": state_bytes, "hip_graph": use_graph},
            "kernels": [{"kernel": p["kernel"], "launches": p["launches"], "ms_per_step": round(p["ms"], 3)} for p in prof[:8]]}


def stream_sweep(args, cfg, weights, device) -> dict:
    """The streaming configuration over a range of stream counts (`--sweep 256,1024,4096`): p50 / p99 of a push per count, and
    the number that matters for serving - how many streams one GPU sustains at real time, i.e. the largest count whose p99 push
    stays inside the audio a push carries (80 ms per frame)."""
    counts = [int(v) for v in args.sweep.split(",")]
    n = args.push_frames
    dec = Decoder(cfg, weights, device=device, precision=args.precision, max_positions=max(8192, (args.steps + args.warmup + 16) * n))
    rows, first = [], None
    for S in counts:
        a = argparse.Namespace(**{**vars(args), "streams": S})
        r = stream_bench(a, cfg, weights, device, dec=dec)
        first = first or r
        rows.append({"streams": S, "p50_ms": r["config"]["p50_ms"], "p99_ms": r["config"]["p99_ms"], "max_ms": r["config"]["max_ms"],
                     "audio_sec_per_sec": r["value"], "real_time_factor_p99": r["config"]["real_time_factor"]})
        torch.cuda.empty_cache()
    budget = n * 80.0
    ok = [r["streams"] for r in rows if r["p99_ms"] <= budget]
    # between the last count inside the budget and the first outside it, p99 grows about linearly with the stream count
    est = None
    for a_, b_ in zip(rows, rows[1:]):
        if a_["p99_ms"] <= budget < b_["p99_ms"]:
            est = int(a_["streams"] + (b_["streams"] - a_["streams"]) * (budget - a_["p99_ms"]) / (b_["p99_ms"] - a_["p99_ms"]))
    out = dict(first)
    out["value"] = rows[-1]["audio_sec_per_sec"]
    out["config"] = {"workload": f"streaming decode sweep: {counts} concurrent streams x {n} frame(s) per push, {args.warmup} warm-up + {args.steps} "
                                 f"timed pushes per count, flagship architecture, random-init weights",
                     "sweep": rows, "push_budget_ms": budget, "largest_count_at_real_time": max(ok) if ok else None,
                     "streams_sustained_at_real_time_estimate": est, "state_bytes_per_stream": first["config"]["state_bytes_per_stream"]}
    out.pop("kernels", None)
    return out


def encode_bench(args, cfg, weights, device) -> dict:
    """The tokenizer's encode side: `--batch` x `--seconds` of 24 kHz audio -> 16 codes per frame (conv encoder, 8-layer
    transformer, stride-2 conv, split-RVQ encode) at the sizes of the sibling's Mimi encoder; the RVQ stage is also timed
    alone on the embeddings.  Algorithmic work per frame: the encoder's contractions + one squared distance to each of the
    2048 rows of 16 code books (3 flops per dimension: subtract, multiply, add)."""
    import importlib
    E = importlib.import_module("qwen3-tts_amd.encoder")
    ecfg = E.EncoderConfig()
    samples = int(round(args.seconds * cfg.sample_rate))
    frames = ecfg.frames(samples)
    w = {**weights, **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=3)}
    dec = Decoder(cfg, w, device=device, precision=args.precision, encoder=ecfg, encoder_max_positions=max(16384, 2 * frames + 8))
    g = torch.Generator().manual_seed(7)
    audio = (torch.randn(args.batch, samples, generator=g) * 0.3).to(device)
    for _ in range(args.warmup):
        codes, emb = dec.encode(audio, return_embeddings=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes = dec.encode(audio)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / args.steps
    assert codes.shape == (args.batch, cfg.num_quantizers, frames) and int(codes.max()) < cfg.codebook_size and int(codes.min()) >= 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dec.rvq_encode(emb)
    torch.cuda.synchronize(device)
    dt_rvq = (time.perf_counter() - t0) / args.steps
    # contraction MACs per second of audio: conv encoder, transformer at 25 Hz, stride-2 conv, input_proj
    macs, dim, rate = ecfg.num_filters * ecfg.kernel_size * cfg.sample_rate, ecfg.num_filters, float(cfg.sample_rate)
    for r in ecfg.ratios:
        hid = dim // ecfg.compress
        macs += rate * (dim * hid * ecfg.residual_kernel_size + hid * dim)
        rate /= r
        macs += rate * (2 * dim) * dim * 2 * r
        dim *= 2
    H, I = ecfg.hidden_size, ecfg.intermediate_size
    macs += rate * (dim * H * ecfg.last_kernel_size + ecfg.num_hidden_layers * (4 * H * H + 2 * H * I))
    rate /= ecfg.downsample_stride
    macs += rate * (H * H * 2 * ecfg.downsample_stride + 2 * cfg.codebook_dim * H)
    rvq_flops = rate * 3.0 * cfg.num_quantizers * cfg.codebook_size * cfg.codebook_dim
    return {"metric": "audio-sec encoded/sec", "value": round(args.batch * args.seconds / dt, 1), "unit": "audio-sec/sec",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"tokenizer encode of {args.batch} x {args.seconds:g} s of 24 kHz audio ({frames} frames each): Mimi-size conv "
                                   f"encoder + transformer + RVQ encode into {cfg.num_quantizers} code books of {cfg.codebook_size} x {cfg.codebook_dim}",
                       "gflop_per_audio_sec": round((2 * macs + rvq_flops) / 1e9, 2),
                       "rvq_stage_ms": round(dt_rvq * 1e3, 3),
                       "rvq_vector_tflops": round(args.batch * args.seconds * rvq_flops / dt_rvq / 1e12, 2)}}


def mixed_bench(args, cfg, weights, device, world, rank, distributed) -> dict:
    """BASELINE.json configs[3]: `--utterances` utterances of 2-30 s (seeded), bucketed by length, LPT-sharded over the
    ranks, decoded with no collective, waveforms gathered to rank 0 over RCCL.  Timed end to end, gather included."""
    import numpy as np
    import torch.distributed as dist
    from qwen3_tts_amd.parallel import decode_sharded, lpt_assign
    rng = np.random.Generator(np.random.PCG64(1234))
    secs = rng.uniform(2.0, 30.0, size=args.utterances)
    frames = [int(round(s * cfg.frame_rate)) for s in secs]
    g = torch.Generator().manual_seed(1234)
    items = [torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, t), generator=g, dtype=torch.int64) for t in frames]
    dec = Decoder(cfg, weights, device=device, precision=args.precision, max_frames_per_launch=args.batch * 325)
    warm = [items[i] for i in range(min(64, len(items)))]
    dec.decode_device(warm)
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    if distributed:
        out = decode_sharded(dec, items, dst=0)
        wavs = out[0] if rank == 0 else None
        torch.cuda.synchronize(device)
        dist.barrier()
    else:
        wavs, _ = dec.decode(items)
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return {}
    assert len(wavs) == len(items) and all(len(w) == cfg.chunked_output_length(t) for w, t in zip(wavs, frames))
    audio = sum(frames) / cfg.frame_rate
    loads = [sum(frames[i] for i in shard) for shard in lpt_assign(frames, world)]
    return {"metric": "audio-sec decoded/sec", "value": round(audio / elapsed, 1), "unit": "audio-sec/sec", "n_gpus": world,
            "steps": 1, "warmup": 1, "ms_per_step": round(elapsed * 1e3, 1), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{len(items)} utterances of 2-30 s (seed 1234), pieces of equal length (padded to buckets of 25 frames, or 10 when those still fill a launch) share launches, LPT-sharded over "
                                   f"{world} rank(s), waveforms gathered to rank 0 and copied to host, flagship architecture",
                       "audio_seconds": round(audio, 1), "rank_load_imbalance": round(max(loads) / (sum(loads) / world), 4)}}


def longform_bench(args, cfg, weights, device, world, rank, distributed) -> dict:
    """BASELINE.json configs[4]: one 10-minute utterance (7500 frames).  (a) the reference's semantics - chunked_decode,
    25 chunks of 300 + 25 frames - with the chunks batched into one launch sequence per GPU (sharded round-robin when
    N > 1); (b) on one GPU also the exact variant: the streaming decode advanced 300 frames per push, state carried
    in HBM, no recomputed context and no seams."""
    import torch.distributed as dist
    from qwen3_tts_amd.parallel import decode_long_sharded
    T = int(round(args.seconds * cfg.frame_rate))
    codes = q3.random_codes(cfg, 1, T, seed=3).to(device)
    dec = Decoder(cfg, weights, device=device, precision=args.precision, max_positions=max(8192, T),
                  max_frames_per_launch=max(args.batch * 325, 26 * 325))
    expect = cfg.chunked_output_length(T)

    def chunked():
        if distributed:
            return decode_long_sharded(dec, codes[0], cfg.chunk_size, cfg.left_context_size, dst=0)
        return dec.chunked_decode(codes, cfg.chunk_size, cfg.left_context_size)

    def timeit(fn, steps, warmup):
        for _ in range(warmup):
            r = fn()
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return r, dt / steps

    wav, dt = timeit(chunked, args.steps, args.warmup)
    extra = {}
    if not distributed:
        assert wav.shape == (1, 1, expect)
        pool = dec.open_streams(1)
        ids = torch.zeros(1, dtype=torch.int32, device=device)

        def exact():
            pool.close_stream(pool.open())            # reset slot 0
            parts = [pool.push_raw(ids, codes[:, :, a:a + 300]) for a in range(0, T, 300)]
            return parts

        parts, dte = timeit(exact, args.steps, args.warmup)
        assert sum(p.shape[1] for p in parts) == T * cfg.total_upsample
        extra = {"exact_streaming_ms": round(dte * 1e3, 2), "exact_streaming_audio_sec_per_sec": round(T / cfg.frame_rate / dte, 1)}
    elif rank == 0:
        assert wav.shape == (expect,)
    if rank != 0:
        return {}
    return {"metric": "audio-sec decoded/sec", "value": round(T / cfg.frame_rate / dt, 1), "unit": "audio-sec/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"one {args.seconds:g} s utterance ({T} frames), chunked decode 300+25 "
                                   f"({len(cfg.chunk_plan(T))} chunks batched, round-robin over {world} rank(s)), flagship architecture",
                       **extra}}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-context-check", action="store_true",
                    help="leave out the extra steps that compute every context row (profiler runs: one kind of step per process)")
    ap.add_argument("--unchunked", action="store_true", help="one forward over all frames instead of chunked_decode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=3, help="utterances in the CPU baseline sample")
    ap.add_argument("--tiny", action="store_true", help="tiny architecture (plumbing check only; not a valid result)")
    ap.add_argument("--workload", default="bulk", choices=["bulk", "stream", "mixed", "longform", "encode"],
                    help="bulk = configs[1] (the headline); stream = configs[2]; mixed = configs[3]; longform = configs[4]")
    ap.add_argument("--utterances", type=int, default=10000)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--push-frames", type=int, default=1)
    ap.add_argument("--sweep", default="", help="stream workload: comma-separated stream counts to measure in one run, e.g. 256,1024,4096")
    ap.add_argument("--no-graph", action="store_true", help="stream workload: plain launches instead of the HIP-graph replay")
    ap.add_argument("--no-stream-record", action="store_true", help="skip the short records of the other configurations (streaming, mixed lengths, long form) appended to the bulk line")
    ap.add_argument("--launch-check", action="store_true", help="every rank prints its rank / world size and exits (no GPU work)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.launch_check:                 # plumbing check of the launcher: no GPU work
        print(json.dumps({"rank": rank, "world": world, "gpus": args.gpus}), flush=True)
        check_world(args.gpus, world)
        return
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()      # the ranks RCCL actually sees
    check_world(args.gpus, world)
    device = torch.device("cuda", local_rank)

    cfg = q3.tiny_config() if args.tiny else q3.flagship_config()
    weights = q3.synthetic_weights(cfg, seed=0)
    if args.workload == "stream":
        # the streaming protocol: 50 warm-up + 1000 timed pushes unless the command line says otherwise
        if "--steps" not in sys.argv:
            args.steps = 1000
        if "--warmup" not in sys.argv:
            args.warmup = 50
        res = stream_sweep(args, cfg, weights, device) if args.sweep else stream_bench(args, cfg, weights, device)
        if distributed:
            # Replicas: every rank runs its own `--streams` streams on its own GPU (a stream never migrates; no collective on
            # the data path).  Node level: the audio of all ranks, and the slowest rank's percentiles.
            import torch.distributed as dist
            mine = [res["value"], res["config"].get("p50_ms", 0.0), res["config"].get("p99_ms", 0.0), res["config"].get("max_ms", 0.0)]
            t = torch.tensor(mine, dtype=torch.float64, device=device)
            allr = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allr, t)
            if rank == 0 and not args.sweep:
                m = torch.stack(allr).cpu()
                res["value"] = round(float(m[:, 0].sum()), 1)
                res["n_gpus"] = world
                res["config"].update({"p50_ms": round(float(m[:, 1].max()), 3), "p99_ms": round(float(m[:, 2].max()), 3),
                                      "max_ms": round(float(m[:, 3].max()), 3), "per_rank_p99_ms": [round(float(v), 3) for v in m[:, 2]],
                                      "placement": f"{world} replicas x {args.streams} streams, one pool per GPU, streams never migrate"})
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(res), flush=True)
        return
    if args.workload == "encode":
        if distributed:
            raise SystemExit("the encode workload is measured on one GPU")
        print(json.dumps(encode_bench(args, cfg, weights, device)), flush=True)
        return
    if args.workload in ("mixed", "longform"):
        if args.workload == "longform" and args.seconds == 30.0:
            args.seconds = 600.0
        fn = mixed_bench if args.workload == "mixed" else longform_bench
        res = fn(args, cfg, weights, device, world, rank, distributed)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    frames = int(round(args.seconds * cfg.frame_rate))
    dec = Decoder(cfg, weights, device=device, precision=args.precision, max_frames_per_launch=args.batch * 325)
    codes = q3.random_codes(cfg, args.batch, frames, seed=1 + rank).to(device)

    def step():
        return dec.forward(codes) if args.unchunked else dec.chunked_decode(codes, cfg.chunk_size, cfg.left_context_size)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        wav = step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(torch.cuda.current_stream(device))
    for _ in range(args.steps):
        wav = step()
    ev1.record(torch.cuda.current_stream(device))
    barrier()
    elapsed = time.perf_counter() - t0
    # Outside the timed region: (a) the per-kernel profile - HIP events around every launch of `psteps` further steps;
    # (b) the same step followed by the copy of its samples to host memory (pinned double buffer), for the
    # PCIe-inclusive rate.
    psteps = min(args.steps, 3)
    dec.profile(True)
    for _ in range(psteps):
        step()
    torch.cuda.synchronize(device)
    prof = dec.profile_read()
    dec.profile(False)
    dec._to_host(wav)                         # allocates the pinned staging buffers
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    for _ in range(psteps):
        host = dec._to_host(step())
    elapsed_d2h = (time.perf_counter() - t1) / psteps
    assert host.shape[0] == wav.numel()
    # (b2) the same utterances through the wrapper the reference calls - `decode(list of code matrices) -> (list of numpy
    # waveforms, sample rate)`: bucketing, the launches, and the copies to host memory running under the later launches
    elapsed_wrapper = None
    if not args.unchunked:
        utts = [codes[b] for b in range(args.batch)]
        dec.decode(utts)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(psteps):
            wavs_host, _sr = dec.decode(utts)
        elapsed_wrapper = (time.perf_counter() - t1) / psteps
        assert len(wavs_host) == args.batch and wavs_host[0].shape[0] == cfg.chunked_output_length(frames)
        assert np.array_equal(wavs_host[0], host[: wavs_host[0].shape[0]]), "the wrapper's samples differ from chunked_decode's"
    # (c) the same step with every row of each chunk's left context computed (Decoder.skip_context off): the chunked decode as a
    # loop of whole forwards would run it.  The samples are bit for bit the same (checked here on the first utterance).
    ms_full_context = None
    if not args.unchunked and not args.no_context_check:
        dec.skip_context = False
        full = step()
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        for _ in range(psteps):
            full = step()
        torch.cuda.synchronize(device)
        ms_full_context = (time.perf_counter() - t2) / psteps * 1e3
        dec.skip_context = True
        assert torch.equal(full[0], wav[0]), "skipping the dead context rows changed the samples"
        del full
    expect = cfg.output_length(frames) if args.unchunked else cfg.chunked_output_length(frames)
    assert wav.shape == (args.batch, 1, expect), wav.shape
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    audio_sec = world * args.batch * frames / cfg.frame_rate * args.steps
    value = audio_sec / elapsed

    out = None
    if rank == 0:
        total_ms = sum(p["ms"] for p in prof) or 1.0
        prof.sort(key=lambda p: -p["ms"])
        dom = prof[0]
        if dom["flops"] > 0:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(dom["kernel"]),
                    "traffic_source": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (profiles/pmc_traffic.json: "
                                      "FETCH_SIZE and WRITE_SIZE in separate passes), not measured in this run",
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "algorithmic_flops_per_launch": round(dom["flops"] / dom["launches"]),
                    "launches_per_step": dom["launches"] // psteps,
                    "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                    "share_of_gpu_time": round(dom["ms"] / total_ms, 3)}
        else:
            achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
        flops_frame = cfg.flops_per_frame()
        whole = value * cfg.frame_rate * flops_frame / 1e12      # algorithmic TFLOP/s of the whole path, all ranks
        out = {
            "metric": "audio-sec decoded/sec", "value": round(value, 1), "unit": "audio-sec/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"batch {args.batch} x {args.seconds:g} s utterances per GPU "
                                   f"({'one un-chunked forward' if args.unchunked else 'chunked decode 300+25'}), "
                                   f"{'tiny' if args.tiny else 'flagship'} architecture, random-init weights, uniform codes",
                       "frames_per_utterance": frames, "gflop_per_audio_sec": round(flops_frame * cfg.frame_rate / 1e9, 2),
                       "whole_path_tflops_per_gpu": round(whole / world, 1),
                       "whole_path_frac_of_bf16_peak": round(whole / world / PEAK_BF16_TFLOPS, 4),
                       "gpu_event_ms_per_step": round(ev0.elapsed_time(ev1) / args.steps, 3),
                       "value_incl_d2h": round(args.batch * frames / cfg.frame_rate / elapsed_d2h, 1),
                       "ms_per_step_incl_d2h": round(elapsed_d2h * 1e3, 3),
                       **({"value_decode_wrapper": round(args.batch * frames / cfg.frame_rate / elapsed_wrapper, 1),
                           "ms_per_step_decode_wrapper": round(elapsed_wrapper * 1e3, 3)} if elapsed_wrapper else {}),
                       **({"dead_context_rows": "the conv-decoder rows under each later chunk's 25 dropped context frames are not "
                                                "computed (q3c_forward_skip); samples bit-identical to whole forwards",
                           "ms_per_step_with_context_rows_computed": round(ms_full_context, 3)} if ms_full_context else {})},
            "roofline": roof,
            "kernels": [{"kernel": p["kernel"], "launches": p["launches"] // psteps, "ms_per_step": round(p["ms"] / psteps, 3),
                         **({"tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 1)} if p["flops"] else
                            {"gbs": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)})} for p in prof[:8]],
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            out["cpu_baseline"] = cpu_baseline(cfg, weights, args.cpu_batch, frames, threads)
        if world == 1 and not args.no_stream_record and not args.tiny:
            # the streaming configuration (BASELINE.json configs[2]) beside the headline, outside its timed region: 256 streams
            # x one frame per push, 100 pushes (a few seconds); the full measurement is `--workload stream`
            sargs = argparse.Namespace(**{**vars(args), "steps": 1000, "warmup": 50})
            sb = stream_bench(sargs, cfg, weights, device)
            out["streaming"] = {"streams": args.streams, "frames_per_push": args.push_frames, "pushes": 1000, "warmup": 50,
                                "p50_ms": sb["config"]["p50_ms"], "p99_ms": sb["config"]["p99_ms"], "audio_sec_per_sec": sb["value"]}
            # short records of BASELINE.json configs[3] and configs[4] as well (same process, after and outside the timed region;
            # `--workload mixed` / `--workload longform` are the full measurements): 2 000 mixed-length utterances end to end
            # including the copy to the host, and one 10-minute utterance chunked 300 + 25
            del dec
            torch.cuda.empty_cache()
            margs = argparse.Namespace(**{**vars(args), "utterances": 2000})
            mb = mixed_bench(margs, cfg, weights, device, 1, 0, False)
            out["mixed_lengths"] = {"utterances": 2000, "audio_seconds": mb["config"]["audio_seconds"], "ms": mb["ms_per_step"],
                                    "audio_sec_per_sec": mb["value"], "includes": "bucketing, launches, copy of every waveform to host memory"}
            torch.cuda.empty_cache()
            largs = argparse.Namespace(**{**vars(args), "seconds": 600.0, "steps": 3, "warmup": 1})
            lb = longform_bench(largs, cfg, weights, device, 1, 0, False)
            out["long_form"] = {"seconds": 600.0, "steps": 3, "warmup": 1, "ms_per_step": lb["ms_per_step"], "audio_sec_per_sec": lb["value"],
                                "exact_streaming_ms": lb["config"].get("exact_streaming_ms")}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
