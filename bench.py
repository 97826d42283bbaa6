#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: real-time factor (audio-sec / wall-sec) and tokens/sec of
Whisper-tiny on 30 s clips, 1/2/4/8 GPUs, weak scaling at 64 utterances per GPU.

One "step" = one full transcribe pass (encoder + prompt prefill + 99 greedy decode steps, "fixed" decode mode of
SURVEY §8d, + the RCCL all-gather of token buffers when N>1) over one batch of synthetic 80x3000 mels that are
already resident in HBM.  Rank 0 prints ONE JSON line.

    python bench.py                                   # N=1: BASELINE config 3 as written — tiny, B=64, bf16 encoder GEMMs; decoder weights,
                                                      # operands and KV cache fp32 (workload tiny_b64_bf16enc_f32dec)
    python bench.py --workload tiny_b64_bf16          # the all-16-bit variant (bf16 operands in the decoder too, bf16 KV)
    python bench.py --workload tiny_b1_f32            # BASELINE config 2
    python bench.py --workload base_b64_f16           # BASELINE config 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # BASELINE config 4 at N=8

Protocol of the timed region: K full passes of 64 clips per GPU, issued in groups of `config.pipeline_depth` = 8 through the
library's pipeline slots (submit the group, collect the group, all-gather the group when N > 1); with `--coalesce 2` (default) the
library pairs consecutive submits into one 128-row decode state — every submit still is B = 64 and receives exactly its own ids.

Besides the contract's keys the line carries (all timed in this same run, N = 1 only):
  value_uncoalesced  the same passes with one decode state per submit, four in flight (round 2's protocol) and the ratio
  value_with_h2d     the same K steps with the 61 MB of mels uploaded from pinned host memory INSIDE the timed region
                     (SURVEY §8d's literal timed region; `value` keeps the mels resident, as the bench contract asks)
  unpipelined        the K steps strictly one after another
  natural            the reference's stop rule (eot 50257, <= 195 iterations, whisper.mojo:205-206) instead of the fixed 99
  value_all_16bit    the same passes with bf16 operands in the decoder too and a bf16 KV cache (round 2's headline; narrower than config 3)
  precision_ladder   the other precisions / configurations, a few steps each: all-16-bit, bf16 operands + fp32 KV, everything fp32
                     (the reference's own precision) at B = 64, BASELINE config 2 (B = 1, fp32) and config 5 (base, B = 64, f16)
  roofline / decode_step / decode_step_4_in_flight / encoder / cpu_baseline   as in round 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The pipelined passes decode on one HIP stream each; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES (default 4)
# hardware queues, and two passes that share a queue run one after the other.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

WORKLOADS = {
    #  name             (config, B/GPU, compute, kv)
    "tiny_b64_bf16": ("tiny", 64, "bf16", "bf16"),
    "tiny_b64_bf16_kv32": ("tiny", 64, "bf16", "f32"),
    "tiny_b64_f32": ("tiny", 64, "f32", "f32"),
    "tiny_b1_f32": ("tiny", 1, "f32", "f32"),
    "base_b64_f16": ("base", 64, "f16", "f16"),
    "tiny_b128_bf16": ("tiny", 128, "bf16", "bf16"),  # two batches of 64 coalesced into one decode state
    # BASELINE config 3 read literally: "bf16 encoder GEMMs" — the decoder's weights, MFMA operands and KV cache stay fp32
    "tiny_b64_bf16enc_f32dec": ("tiny", 64, "bf16", "f32"),
}
DECODER_FP32 = {"tiny_b64_bf16enc_f32dec"}
DEFAULT_WORKLOAD = "tiny_b64_bf16enc_f32dec"  # BASELINE config 3 as written
LADDER = ["tiny_b64_bf16enc_f32dec", "tiny_b64_bf16", "base_b64_f16", "tiny_b64_bf16_kv32", "tiny_b64_f32", "tiny_b1_f32", "tiny_b128_bf16"]
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
DECODE_STEPS = 99      # + 1 token from the prefill = 100 generated ids per utterance
NATURAL_LOOP = 195     # whisper.mojo:205
CLIP_SECONDS = 30.0


def precision_of(workload, cdt, kdt):
    """(dtype field, sentence) — what is narrowed below the reference's fp32 and what is not."""
    if workload in DECODER_FP32:
        return (f"{cdt} encoder GEMMs; f32 decoder + KV",
                f"GEMM operands {cdt} in the ENCODER only (conv stem, encoder blocks, cross-K/V projection: weights and the activations fed to "
                f"MFMA); the decoder's weights and MFMA operands and the self-/cross-attention KV cache are fp32 — BASELINE config 3 as written")
    if cdt == "f32":
        return ("f32", "everything fp32 (exact-fp32 MFMA), the reference's precision")
    return (f"{cdt} operands, {kdt} KV",
            f"GEMM operands {cdt} in the encoder AND the decoder (weights, activations fed to MFMA), KV cache {kdt}")


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, weights, mel, max_loop):
    """The CPU restatement of the reference (oracle/, OpenMP like the reference's `parallelize`) on ONE clip of the same
    workload.  Picks the fastest thread count among {8,16,32} (1 run each; all-core teams oversubscribe the box's CPU quota and stall), then 1 warm-up + median of 3.
    Checker code — timed here as the reported baseline, never used by the product."""
    import ctypes
    from oracle import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gomp = ctypes.CDLL("libgomp.so.1")
    ref = oracle.OracleModel(cfg, weights, gelu_mode=0)

    def run():
        t0 = time.perf_counter()
        ref.transcribe(mel=mel, max_loop=max_loop, ignore_eot=True)
        return time.perf_counter() - t0

    best_n, best_t = None, None
    for n in sorted({min(avail, k) for k in (8, 16, 32)}):
        gomp.omp_set_num_threads(n)
        t = run()
        log(f"  oracle with {n} threads: {t:.2f} s/clip")
        if best_t is None or t < best_t:
            best_n, best_t = n, t
    gomp.omp_set_num_threads(best_n)
    ts = [run() for _ in range(4)]
    t = float(np.median(ts[1:]))
    return {"value": round(CLIP_SECONDS / t, 2), "unit": "x real-time", "cores": best_n, "kind": "port",
            "host_cores_available": avail, "seconds_per_clip": round(t, 3), "tokens_per_sec": round((max_loop + 1) / t, 1),
            "sample": f"1 clip of the same workload (fp32, 1 prefill + {max_loop} decode steps), 1 warm-up + median of 3 at the "
                      "fastest of {8,16,32} OpenMP threads; C restatement of the reference (Mojo toolchain absent)"}


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r*_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 FETCH correction applied).  Counters need
    their own serialised profiler passes, so they cannot be collected inside a timed bench run; null for other workloads."""
    files = {"tiny_b64_bf16enc_f32dec": ("r3_pmc_traffic.json",), "tiny_b64_bf16": ("r2_pmc_traffic.json", "r1_pmc_traffic.json")}
    for name in files.get(workload, ()):
        path = os.path.join(ROOT, "profiles", name)
        try:
            ks = json.load(open(path))["kernels"]
            return next(v["traffic_bytes"] for k, v in ks.items() if k.startswith("attn_decode_kernel"))
        except Exception:
            continue
    return None


class Bench:
    """One loaded model + its shard of synthetic mels, with the timing legs."""

    def __init__(self, workload, batch, rank, world, local, weights_cache, coalesce=0):
        import ctypes as C
        import torch
        from whisper_mojo_amd import DT_BF16, DT_F16, DT_F32, WhisperConfig, _lib, dist as wdist
        from whisper_mojo_amd.loader import WeightLoader
        from whisper_mojo_amd.whisper import Whisper
        self.C, self.torch, self._lib, self.wdist = C, torch, _lib, wdist
        self.workload, self.world, self.rank = workload, world, rank
        cfg_name, B, self.cdt, self.kdt = WORKLOADS[workload]
        self.B = batch or B
        self.cfg_name = cfg_name
        self.cfg = WhisperConfig.tiny() if cfg_name == "tiny" else WhisperConfig.base()
        DT = {"f32": DT_F32, "bf16": DT_BF16, "f16": DT_F16}
        self.dev = torch.device("cuda", local)
        self.L = _lib.lib()
        if cfg_name not in weights_cache:  # synthetic weights (seed 0) in the reference's flat file format, by the library's C generator
            dims = self.cfg.dims()
            w = np.empty(self.cfg.weight_count(), np.float32)
            self.L.wm_synth_weights(C.byref(dims), 0, w.ctypes.data_as(C.POINTER(C.c_float)))
            weights_cache[cfg_name] = w
        self.weights = weights_cache[cfg_name]
        self.coalesce = coalesce
        self.model = Whisper(self.cfg, compute_dtype=DT[self.cdt], kv_dtype=DT[self.kdt], max_batch=self.B, device=local,
                             decoder_fp32=workload in DECODER_FP32, coalesce=coalesce)
        self.model.load(WeightLoader.from_array(self.weights))
        # this rank's shard of the global batch: utterance u uses mel seed 1000+u
        self.total = self.B * world
        self.first, self.count = wdist.shard_range(self.total, rank, world)
        # pinned host staging (the H2D-inclusive leg uploads from here) and the resident copy
        self.mel_pinned = torch.empty((self.count, self.cfg.n_mels, self.cfg.n_frames), dtype=torch.float32).pin_memory()
        self.mel_host = self.mel_pinned.numpy()
        for i in range(self.count):
            self.L.wm_synth_mel_host(1000 + self.first + i, self.cfg.n_mels, self.cfg.n_frames,
                                     self.mel_host[i].ctypes.data_as(C.POINTER(C.c_float)))
        self.mel_dev = self.mel_pinned.to(self.dev)  # resident in HBM before the timed region
        self.last_counts = []
        # N > 1 over RCCL: the ids stay on the device as the gather buffer (wm_transcribe_wait_device) — no host round trip before
        # the all-gather.  (gloo rehearsals and N = 1 take the host path; WM_BENCH_DEVICE_GATHER=1 forces the device path at N = 1.)
        dg = os.environ.get("WM_BENCH_DEVICE_GATHER")
        self.device_gather = bool(dg) or (world > 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl")
        self.rows_per_rank = (self.total + world - 1) // world

    def close(self):
        self.model.close()

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.torch.distributed.barrier()
            self.torch.cuda.synchronize()

    def run_steps(self, n, depth, max_loop=DECODE_STEPS, ignore_eot=True, mel=None, gather=True, eot=None):
        """n full passes, `depth` of them in flight through the library's pipeline slots (depth 1: wm_transcribe, synchronous).
        Every pass does the full work; all n are complete (ids on the host, gathered) when this returns."""
        m, mel = self.model, (self.mel_dev if mel is None else mel)
        stride = 4 + 1 + max_loop
        out = None
        kw = {} if eot is None else {"eot": eot}
        if depth <= 1:
            for _ in range(n):
                m.transcribe_batch(mel, max_loop=max_loop, ignore_eot=ignore_eot, **kw)
                self.last_counts = [m.last_counts]
                if gather:
                    out = self.wdist.gather_tokens(m.last_tokens, m.last_counts, self.total, stride)
            return out
        # Groups of `depth` passes: submit them all, collect them all, THEN all-gather.  Four passes on four streams fill
        # the four hardware queues the chip runs at a time and finish together anyway; an RCCL kernel issued while a slot's
        # queue is busy could land on that queue's pipe and sit behind a whole pass, so the gathers go where the GPU is idle.
        k = 0
        while k < n:
            g = min(depth, n - k)
            for sl in range(g):
                m.transcribe_submit(mel, slot=sl, max_loop=max_loop, ignore_eot=ignore_eot, **kw)
            done = []
            if gather and self.device_gather:
                for sl in range(g):
                    packed = self.torch.empty((self.rows_per_rank, 1 + stride), dtype=self.torch.int32, device=self.dev)
                    m.transcribe_wait_device(sl, packed)
                    done.append(packed)
                for packed in done:
                    out = self.wdist.gather_tokens_device(packed, self.total)
                self.last_counts = [np.asarray([len(o) for o in out[self.first:self.first + self.count]], np.int32)]
                k += g
                continue
            for sl in range(g):
                m.transcribe_wait(sl)
                done.append((m.last_tokens, m.last_counts))
            if gather:
                for toks, cnts in done:
                    out = self.wdist.gather_tokens(toks, cnts, self.total, stride)
            self.last_counts = [c for _, c in done]
            k += g
        return out

    def timed(self, n, depth, **kw):
        """(seconds, max over ranks; last gathered ids) for n passes bracketed by barrier + synchronize on both sides."""
        self.sync()
        t0 = time.perf_counter()
        out = self.run_steps(n, depth, **kw)
        self.sync()
        dt = time.perf_counter() - t0
        if self.world > 1:
            red = self.dev if self.torch.distributed.get_backend() == "nccl" else self.torch.device("cpu")
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=red)
            self.torch.distributed.all_reduce(t, op=self.torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    def setup_slots(self, depth, max_loop=2):
        """allocate every slot's state and capture its step graph (set-up, like loading the weights — not a step)"""
        if depth > 1:
            for sl in range(depth):
                self.model.transcribe_submit(self.mel_dev, slot=sl, max_loop=max_loop, ignore_eot=True)
            for sl in range(depth):
                self.model.transcribe_wait(sl)
        else:
            self.model.transcribe_batch(self.mel_dev, max_loop=max_loop, ignore_eot=True)
        self.sync()

    def rates(self, dt, n, gen_per_utt):
        return (self.total * CLIP_SECONDS * n / dt, self.total * gen_per_utt * n / dt)

    def kernel_timings(self, x4=True):
        """HIP-event timings on the library's own streams (wm_bench_kernel): dominant kernel, decode step, encoder."""
        C, L, _lib, m = self.C, self.L, self._lib, self.model
        st = C.c_void_p()
        _lib.check(L.wm_state_new(m._h, self.count, C.byref(st)))
        _lib.check(L.wm_encode(m._h, st, C.c_void_p(self.mel_dev.data_ptr()), 1, self.count, None))
        us, nbytes, step_us, step_bytes, enc_us = C.c_float(), C.c_double(), C.c_float(), C.c_double(), C.c_float()
        _lib.check(L.wm_bench_kernel(m._h, st, _lib.KERNEL_CROSS_ATTN, 200, C.byref(us)))
        _lib.check(L.wm_bench_bytes(m._h, st, _lib.KERNEL_CROSS_ATTN, C.byref(nbytes)))
        _lib.check(L.wm_bench_kernel(m._h, st, _lib.KERNEL_DECODE_STEP, 50, C.byref(step_us)))
        _lib.check(L.wm_bench_bytes(m._h, st, _lib.KERNEL_DECODE_STEP, C.byref(step_bytes)))
        step4_us = None
        if x4:
            # the same step with four passes decoding at once (what the pipelined rate runs on): four states, four host threads
            import threading
            sts = [st]
            for _ in range(3):
                s2 = C.c_void_p()
                _lib.check(L.wm_state_new(m._h, self.count, C.byref(s2)))
                _lib.check(L.wm_encode(m._h, s2, C.c_void_p(self.mel_dev.data_ptr()), 1, self.count, None))
                sts.append(s2)
            us4 = [C.c_float() for _ in sts]
            errs = []

            def _chain(i):
                try:
                    _lib.check(L.wm_bench_kernel(m._h, sts[i], _lib.KERNEL_DECODE_STEP_SHARED, 100, C.byref(us4[i])))
                except Exception as e:  # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=_chain, args=(i,)) for i in range(4)]
            [t.start() for t in th]
            [t.join() for t in th]
            if errs:
                raise errs[0]
            step4_us = max(u.value for u in us4)
            for s2 in sts[1:]:
                L.wm_state_free(s2)
        _lib.check(L.wm_bench_kernel(m._h, st, _lib.KERNEL_ENCODER, 3, C.byref(enc_us)))
        L.wm_state_free(st)
        cfg = self.cfg
        d, f, Lr, T = cfg.d_model, cfg.ffn, cfg.n_layers, cfg.n_audio_ctx
        enc_flops = self.count * (2.0 * 3000 * d * cfg.n_mels * 3 + 2.0 * T * d * d * 3 +
                                  Lr * (2.0 * T * d * d * 4 + 4.0 * T * T * d + 4.0 * T * d * f) + 2.0 * T * d * d * 2 * Lr)
        achieved = nbytes.value / (us.value * 1e-6) / 1e9
        step_gbs = step_bytes.value / (step_us.value * 1e-6) / 1e9
        res = {
            "roofline": {"bound": "hbm", "kernel": "attn_decode_kernel (decoder cross-attention, one layer, all utterances)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(self.workload),
                         "bytes_per_launch": nbytes.value, "us_per_launch": round(us.value, 2)},
            "decode_step": {"us": round(step_us.value, 1), "algorithmic_bytes": step_bytes.value,
                            "GBps": round(step_gbs, 1), "frac_of_hbm_peak": round(step_gbs / HBM_PEAK_GBS, 4)},
            "encoder": {"ms": round(enc_us.value / 1e3, 3), "TFLOPs": round(enc_flops / (enc_us.value * 1e-6) / 1e12, 1)},
        }
        if step4_us is not None:
            agg = 4 * step_bytes.value / (step4_us * 1e-6) / 1e9
            res["decode_step_4_in_flight"] = {"us_per_step_of_each_chain": round(step4_us, 1), "aggregate_GBps": round(agg, 1),
                                              "frac_of_hbm_peak": round(agg / HBM_PEAK_GBS, 4)}
        return res


def ladder_entry(name, rank, world, local, weights_cache, depth):
    """A few pipelined steps + the decode-step timing of another precision / batch of the same model (N = 1 leg)."""
    b = Bench(name, 0, rank, world, local, weights_cache)
    try:
        d = min(depth, 4)
        b.setup_slots(d)
        b.run_steps(d, d)
        n = 2 * d
        dt, _ = b.timed(n, d)
        b.run_steps(1, 1)
        seq, _ = b.timed(2, 1)
        k = b.kernel_timings(x4=False)
        rtf, tok = b.rates(dt, n, DECODE_STEPS + 1)
        co = None
        if b.B == 64:  # the same passes coalesced in pairs by the library, eight submits in flight (the headline's protocol)
            b.close()
            b = Bench(name, 0, rank, world, local, weights_cache, coalesce=2)
            b.setup_slots(8)
            b.run_steps(8, 8)
            c_dt, _ = b.timed(16, 8)
            co = {"value": round(b.rates(c_dt, 16, 1)[0], 1), "ms_per_step": round(c_dt / 16 * 1e3, 3), "steps": 16, "pipeline_depth": 8}
        return {"workload": name, "model": b.cfg_name, "coalesced": co, "operands": b.cdt + (" (encoder only; decoder fp32)" if name in DECODER_FP32 else ""), "kv": b.kdt, "utterances": b.B, "steps": n, "ms_per_step": round(dt / n * 1e3, 3),
                "value": round(rtf, 1), "tokens_per_sec": round(tok, 1), "unpipelined_ms_per_step": round(seq / 2 * 1e3, 3),
                "decode_step": k["decode_step"], "cross_attention": {k2: k["roofline"][k2] for k2 in ("achieved", "frac", "us_per_launch")},
                "encoder": k["encoder"]}
    finally:
        b.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)  # a multiple of the pipeline depth: passes complete in groups of eight (four uncoalesced)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-x4", action="store_true", help="skip the four-chains-in-flight decode step timing (profiler runs: keeps every launch of the dominant kernel alone on the chip)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8],
                    help="steps in flight (library pipeline slots); 0 = 8 with --coalesce 2 (four 128-row passes), else 4")
    ap.add_argument("--coalesce", type=int, default=2, choices=[0, 2],
                    help="2: the library pairs consecutive 64-clip submits into one 128-row decode state (wm_config.coalesce); every "
                         "submit still is B = 64 and gets its own ids.  0: one decode state per submit")
    ap.add_argument("--no-extras", action="store_true", help="skip the value_with_h2d / natural / precision_ladder legs (profiler and test runs)")
    ap.add_argument("--dump-ids", default="", help="rank 0 writes the gathered ids of the last timed step to this .npy file (tests)")
    args = ap.parse_args()

    import torch
    from whisper_mojo_amd import dist as wdist

    # rehearsal knobs (not used by the driver): WM_BENCH_BACKEND=gloo + WM_BENCH_SINGLE_DEVICE=1 run N ranks on ONE GPU
    backend = os.environ.get("WM_BENCH_BACKEND", "nccl")
    rank, local, world = wdist.init_from_env(backend)
    if os.environ.get("WM_BENCH_DEVICE_GATHER") and world == 1 and not torch.distributed.is_initialized():
        # one-GPU rehearsal of the RCCL path: a world of one still initialises RCCL and runs the device-buffer gather code
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(0)
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    if os.environ.get("WM_BENCH_SINGLE_DEVICE"):
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    if args.no_pipeline:
        args.coalesce = 0
    depth = 1 if args.no_pipeline else (args.pipeline or (8 if args.coalesce == 2 else 4))
    weights_cache = {}
    log("generating weights, loading model, synthesising mels")
    b = Bench(args.workload, args.batch, rank, world, local, weights_cache, coalesce=args.coalesce)
    stride = 4 + 1 + DECODE_STEPS

    log("set-up of the pipeline slots; warm-up")
    b.setup_slots(depth)
    b.run_steps(args.warmup, depth)
    log("timed region")
    dt, out = b.timed(args.steps, depth)
    assert out is not None and len(out) == b.total and all(len(o) == stride for o in out)
    log(f"timed region done: {dt:.3f} s")
    if args.dump_ids and rank == 0:
        np.save(args.dump_ids, np.asarray(out, np.int32))

    extras = {}
    if depth > 1:  # the same K steps strictly one after another (no overlap between steps), for reference
        b.run_steps(1, 1)
        seq_dt, _ = b.timed(args.steps, 1)
        extras["unpipelined"] = {"ms_per_step": round(seq_dt / args.steps * 1e3, 3), "value": round(b.rates(seq_dt, args.steps, 1)[0], 1)}
    if world == 1:
        # the decode step as a pass really runs it (cache length growing 4 .. 103): synchronous passes with and without the loop
        b.run_steps(1, 1, max_loop=0)
        t_full, _ = b.timed(3, 1, gather=False)
        t_none, _ = b.timed(3, 1, max_loop=0, gather=False)
        extras["_in_pass_step_us"] = (t_full - t_none) / 3 / DECODE_STEPS * 1e6
    if world == 1 and not args.no_extras:
        log("H2D-inclusive leg")
        b.run_steps(min(depth, args.steps), depth, mel=b.mel_host)
        h_dt, _ = b.timed(args.steps, depth, mel=b.mel_host)  # pinned host mels: every pass uploads its 61 MB inside the timed region
        extras["value_with_h2d"] = {"value": round(b.rates(h_dt, args.steps, 1)[0], 1), "ms_per_step": round(h_dt / args.steps * 1e3, 3),
                                    "note": "mels start in pinned host memory; each pass's hipMemcpyAsync is inside the timed region"}
        log("natural decode mode (reference stop rule)")
        nat_n = max(depth, 4)
        b.setup_slots(depth, max_loop=NATURAL_LOOP)
        n_dt, _ = b.timed(nat_n, depth, max_loop=NATURAL_LOOP, ignore_eot=False)
        gen = float(np.mean([np.mean(c) for c in b.last_counts])) - 4
        b.run_steps(1, 1, max_loop=NATURAL_LOOP, ignore_eot=False)
        s_dt, _ = b.timed(2, 1, max_loop=NATURAL_LOOP, ignore_eot=False)  # synchronous form: polls "all finished" and stops early
        extras["natural"] = {"stop_rule": "eot 50257 or 195 loop iterations (whisper.mojo:205-206)", "steps": nat_n,
                             "ms_per_step": round(n_dt / nat_n * 1e3, 3), "value": round(b.rates(n_dt, nat_n, 1)[0], 1),
                             "generated_ids_per_utterance": round(gen, 1), "tokens_per_sec": round(b.total * gen * nat_n / n_dt, 1),
                             "synchronous_ms_per_step": round(s_dt / 2 * 1e3, 3),
                             "note": "random-init weights almost never emit eot: natural mode runs to the 195-iteration bound"}
        # the same stop rule with an eot every utterance reaches: 64 copies of clip 0, eot := the id it emits at iteration 60 — the
        # loop is cut there (whisper.mojo:206-207), at most two sub-chunks of 8 steps late
        same = b.mel_dev[:1].expand(b.count, -1, -1).contiguous()
        free = b.model.transcribe_batch(same[:1], max_loop=NATURAL_LOOP, ignore_eot=True)[0]
        eot = int(free[4 + 60])
        stop = free.index(eot, 4) - 4  # loop iterations until the first occurrence
        b.run_steps(min(depth, 2), depth, max_loop=NATURAL_LOOP, ignore_eot=False, mel=same, eot=eot, gather=False)
        r_dt, _ = b.timed(nat_n, depth, max_loop=NATURAL_LOOP, ignore_eot=False, mel=same, eot=eot, gather=False)
        steps_run = b.model.loop_steps(0)
        b.run_steps(1, 1, max_loop=NATURAL_LOOP, ignore_eot=False, mel=same, eot=eot, gather=False)
        rs_dt, _ = b.timed(2, 1, max_loop=NATURAL_LOOP, ignore_eot=False, mel=same, eot=eot, gather=False)
        extras["natural_reachable_eot"] = {"what": "64 copies of clip 0, eot := the id it emits at loop iteration 60", "eot_at_iteration": stop,
                                           "loop_iterations_enqueued": steps_run, "steps": nat_n, "ms_per_step": round(r_dt / nat_n * 1e3, 3),
                                           "synchronous_ms_per_step": round(rs_dt / 2 * 1e3, 3), "synchronous_loop_iterations": b.model.loop_steps(0)}
        del same
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        rtf, tok_s = b.rates(dt, args.steps, DECODE_STEPS + 1)
        log("kernel timings (cross-attention, decode step, encoder)")
        k = b.kernel_timings(x4=not args.no_x4)
        dtype, prec = precision_of(args.workload, b.cdt, b.kdt)
        res = {
            "metric": "real-time-factor (audio-sec/wall-sec)", "value": round(rtf, 1), "unit": "x real-time",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "tokens_per_sec": round(tok_s, 1),
            "config": {"workload": f"whisper-{b.cfg_name}, {b.B} synthetic 80x3000 mels per GPU ({b.total} total) resident in HBM, greedy, "
                                   f"1 prefill + {DECODE_STEPS} decode steps; {prec}; accumulation / LayerNorm / softmax / residual fp32; "
                                   "random-init weights (seed 0)",
                       "name": args.workload, "utterances_per_gpu": b.B, "kv_dtype": b.kdt, "parallelism": f"dp{world}",
                       "pipeline_depth": depth,
                       "coalesce": ("the library pairs consecutive 64-clip submits into one 128-row decode state (wm_config.coalesce = 2): "
                                    f"{depth} submits in flight = {depth // 2} passes of 128 rows; every submit is B = 64 and receives its own ids, "
                                    "bit-identical to the uncoalesced run (tests/test_gpu_edges.py)") if args.coalesce == 2 else "off"},
        }
        in_pass = extras.pop("_in_pass_step_us", None)
        res.update(extras)
        res.update(k)
        if in_pass is not None:  # next to the replayed step (cache length held at 50): the average over a real pass's 99 steps
            sb = res["decode_step"]["algorithmic_bytes"]
            res["decode_step"]["cache_len"] = 50
            res["decode_step"]["in_pass_avg_us"] = round(in_pass, 1)
            res["decode_step"]["in_pass_frac_of_hbm_peak"] = round(sb / (in_pass * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if world == 1 and not args.no_extras:
            if args.coalesce == 2:  # the same workload with one decode state per submit, four in flight (round 2's protocol)
                log("uncoalesced comparison")
                b.model.close()
                u = Bench(args.workload, args.batch, rank, world, local, weights_cache, coalesce=0)
                try:
                    u.setup_slots(4)
                    u.run_steps(4, 4)
                    u_dt, _ = u.timed(12, 4)
                    res["value_uncoalesced"] = {"value": round(u.rates(u_dt, 12, 1)[0], 1), "ms_per_step": round(u_dt / 12 * 1e3, 3), "steps": 12,
                                                "pipeline_depth": 4, "ratio": round(res["value"] / u.rates(u_dt, 12, 1)[0], 4)}
                finally:
                    u.close()
            res["precision_ladder"] = []
            for name in LADDER:
                if name == args.workload:
                    continue
                log(f"precision ladder: {name}")
                b.model.close()  # one model resident at a time
                res["precision_ladder"].append(ladder_entry(name, rank, world, local, weights_cache, depth))
                if name == "tiny_b64_bf16":  # round 2's headline, kept as a named extra: NOT config 3 (decoder and KV narrowed too)
                    e = res["precision_ladder"][-1]
                    res["value_all_16bit"] = {"value": e["value"], "ms_per_step": e["ms_per_step"], "steps": e["steps"], "coalesced": e["coalesced"],
                                              "note": "bf16 operands in the decoder too + bf16 KV cache: narrower than BASELINE config 3"}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle) ...")
            res["cpu_baseline"] = cpu_baseline(b.cfg, b.weights, b.mel_host[0], DECODE_STEPS)
        print(json.dumps(res), flush=True)
    b.close()
    if world > 1 or torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
