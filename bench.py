#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: real-time factor (audio-sec / wall-sec) and tokens/sec of
Whisper-tiny on 30 s clips, 1/2/4/8 GPUs, weak scaling at 64 utterances per GPU.

One "step" = one full transcribe pass (encoder + prompt prefill + 99 greedy decode steps, "fixed" decode mode of
SURVEY §8d, + the RCCL all-gather of token buffers when N>1) over one batch of synthetic 80x3000 mels that are
already resident in HBM.  Rank 0 prints ONE JSON line.

    python bench.py                                   # N=1, tiny, B=64, bf16 operands (BASELINE config 3)
    python bench.py --workload tiny_b1_f32            # BASELINE config 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # BASELINE config 4 at N=8
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
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
DECODE_STEPS = 99      # + 1 token from the prefill = 100 generated ids per utterance
CLIP_SECONDS = 30.0


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
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r1_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 x2 FETCH correction applied).  Counters need
    their own serialised profiler passes, so they cannot be collected inside a timed bench run; null for other workloads."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    if workload != "tiny_b64_bf16" or not os.path.exists(path):
        return None
    try:
        ks = json.load(open(path))["kernels"]
        return next(v["traffic_bytes"] for k, v in ks.items() if k.startswith("attn_decode_kernel"))
    except Exception:
        return None


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)  # a multiple of the pipeline depth: passes complete in groups of four
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="tiny_b64_bf16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-x4", action="store_true", help="skip the four-chains-in-flight decode step timing (profiler runs: keeps every launch of the dominant kernel alone on the chip)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another")
    ap.add_argument("--pipeline", type=int, default=4, choices=[1, 2, 3, 4, 5, 6, 7, 8], help="steps in flight (library pipeline slots)")
    ap.add_argument("--dump-ids", default="", help="rank 0 writes the gathered ids of the last timed step to this .npy file (tests)")
    args = ap.parse_args()

    import torch
    from whisper_mojo_amd import DT_BF16, DT_F16, DT_F32, WhisperConfig, _lib, dist as wdist
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    import ctypes as C

    # rehearsal knobs (not used by the driver): WM_BENCH_BACKEND=gloo + WM_BENCH_SINGLE_DEVICE=1 run N ranks on ONE GPU
    backend = os.environ.get("WM_BENCH_BACKEND", "nccl")
    rank, local, world = wdist.init_from_env(backend)
    if os.environ.get("WM_BENCH_SINGLE_DEVICE"):
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cfg_name, B, cdt, kdt = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    cfg = WhisperConfig.tiny() if cfg_name == "tiny" else WhisperConfig.base()
    DT = {"f32": DT_F32, "bf16": DT_BF16, "f16": DT_F16}
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # synthetic weights (seed 0) in the reference's flat file format, generated by the library's C generator
    L = _lib.lib()
    dims = cfg.dims()
    weights = np.empty(cfg.weight_count(), np.float32)
    L.wm_synth_weights(C.byref(dims), 0, weights.ctypes.data_as(C.POINTER(C.c_float)))
    log(f"weights generated ({weights.nbytes} bytes); loading model")
    model = Whisper(cfg, compute_dtype=DT[cdt], kv_dtype=DT[kdt], max_batch=B, device=local)
    model.load(WeightLoader.from_array(weights))

    # this rank's shard of the global batch: utterance u uses mel seed 1000+u
    total = B * world
    first, count = wdist.shard_range(total, rank, world)
    mel_host = np.empty((count, cfg.n_mels, cfg.n_frames), np.float32)
    for i in range(count):
        L.wm_synth_mel_host(1000 + first + i, cfg.n_mels, cfg.n_frames,
                            mel_host[i].ctypes.data_as(C.POINTER(C.c_float)))
    mel_dev = torch.from_numpy(mel_host).to(dev)  # resident in HBM before the timed region
    stride = 4 + 1 + DECODE_STEPS

    # Steps are issued through the library's pipeline slots (wm_transcribe_submit / _wait), `--pipeline` of them in flight:
    # their encoders share the chip, then their latency/HBM-bound decode chains do.  Every step still does the full work;
    # all K steps are complete (ids on the host, gathered) before the timed region closes.  --no-pipeline runs them one
    # after another.
    def run_steps(n):
        out = None
        if args.no_pipeline:
            for _ in range(n):
                model.transcribe_batch(mel_dev, max_loop=DECODE_STEPS, ignore_eot=True)
                out = wdist.gather_tokens(model.last_tokens, model.last_counts, total, stride)
            return out
        # Groups of `depth` passes: submit them all, collect them all, THEN all-gather.  Four passes on four streams fill
        # the four hardware queues the chip runs at a time and finish together anyway; an RCCL kernel issued while a slot's
        # queue is busy could land on that queue's pipe and sit behind a whole pass, so the gathers go where the GPU is idle.
        depth, k = args.pipeline, 0
        while k < n:
            g = min(depth, n - k)
            for sl in range(g):
                model.transcribe_submit(mel_dev, slot=sl, max_loop=DECODE_STEPS, ignore_eot=True)
            done = []
            for sl in range(g):
                model.transcribe_wait(sl)
                done.append((model.last_tokens, model.last_counts))
            for toks, cnts in done:
                out = wdist.gather_tokens(toks, cnts, total, stride)
            k += g
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    log("model loaded, mels resident; set-up of the pipeline slots")
    if not args.no_pipeline:  # allocate every slot's state and capture its step graph (set-up, not a step)
        for sl in range(args.pipeline):
            model.transcribe_submit(mel_dev, slot=sl, max_loop=2, ignore_eot=True)
        for sl in range(args.pipeline):
            model.transcribe_wait(sl)
    sync()
    log("warm-up")
    out = run_steps(args.warmup)
    sync()
    log("timed region")
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert out is not None and len(out) == total and all(len(o) == stride for o in out)
    if args.dump_ids and rank == 0:
        np.save(args.dump_ids, np.asarray(out, np.int32))

    log(f"timed region done: {dt:.3f} s")
    # the same K steps strictly one after another (no overlap between steps), for reference
    seq_dt = None
    if not args.no_pipeline and args.pipeline > 1:
        saved, args.no_pipeline = args.no_pipeline, True
        run_steps(1)
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps)
        sync()
        seq_dt = time.perf_counter() - t1
        args.no_pipeline = saved
        if world > 1:
            t = torch.tensor([seq_dt], dtype=torch.float64, device=red_dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            seq_dt = float(t.item())
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        rtf = total * CLIP_SECONDS * args.steps / dt
        tok_s = total * (DECODE_STEPS + 1) * args.steps / dt
        # dominant kernel = decode cross-attention (cross-K/V streaming); timed with HIP events on the library's stream
        st = C.c_void_p()
        _lib.check(L.wm_state_new(model._h, count, C.byref(st)))
        _lib.check(L.wm_encode(model._h, st, C.c_void_p(mel_dev.data_ptr()), 1, count, None))
        us, nbytes, step_us, step_bytes, enc_us = C.c_float(), C.c_double(), C.c_float(), C.c_double(), C.c_float()
        log("kernel timing: cross-attention")
        _lib.check(L.wm_bench_kernel(model._h, st, _lib.KERNEL_CROSS_ATTN, 200, C.byref(us)))
        _lib.check(L.wm_bench_bytes(model._h, st, _lib.KERNEL_CROSS_ATTN, C.byref(nbytes)))
        log("kernel timing: decode step")
        _lib.check(L.wm_bench_kernel(model._h, st, _lib.KERNEL_DECODE_STEP, 50, C.byref(step_us)))
        _lib.check(L.wm_bench_bytes(model._h, st, _lib.KERNEL_DECODE_STEP, C.byref(step_bytes)))
        step4_us = None
        if not args.no_x4:
            # the same step with four passes decoding at once (what the pipelined rate runs on): four states, four host threads
            log("kernel timing: decode step, four chains in flight")
            import threading
            sts = [st]
            for _ in range(3):
                s2 = C.c_void_p()
                _lib.check(L.wm_state_new(model._h, count, C.byref(s2)))
                _lib.check(L.wm_encode(model._h, s2, C.c_void_p(mel_dev.data_ptr()), 1, count, None))
                sts.append(s2)
            us4 = [C.c_float() for _ in sts]
            errs = []

            def _chain(i):
                try:
                    _lib.check(L.wm_bench_kernel(model._h, sts[i], _lib.KERNEL_DECODE_STEP, 100, C.byref(us4[i])))
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
        log("kernel timing: encoder")
        _lib.check(L.wm_bench_kernel(model._h, st, _lib.KERNEL_ENCODER, 3, C.byref(enc_us)))
        L.wm_state_free(st)
        achieved = nbytes.value / (us.value * 1e-6) / 1e9
        step_gbs = step_bytes.value / (step_us.value * 1e-6) / 1e9
        d, f, Lr, T = cfg.d_model, cfg.ffn, cfg.n_layers, cfg.n_audio_ctx
        enc_flops = count * (2.0 * 3000 * d * cfg.n_mels * 3 + 2.0 * T * d * d * 3 +
                             Lr * (2.0 * T * d * d * 4 + 4.0 * T * T * d + 4.0 * T * d * f) + 2.0 * T * d * d * 2 * Lr)
        res = {
            "metric": "real-time-factor (audio-sec/wall-sec)", "value": round(rtf, 1), "unit": "x real-time",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cdt, "data": "synthetic",
            "tokens_per_sec": round(tok_s, 1),
            "unpipelined": None if seq_dt is None else {"ms_per_step": round(seq_dt / args.steps * 1e3, 3),
                                                        "value": round(total * CLIP_SECONDS * args.steps / seq_dt, 1)},
            "config": {"workload": f"whisper-{cfg_name}, {B} synthetic 80x3000 mels per GPU ({total} total), greedy, 1 prefill + "
                                   f"{DECODE_STEPS} decode steps, operands {cdt}, KV cache {kdt}, random-init weights (seed 0)",
                       "name": args.workload, "utterances_per_gpu": B, "kv_dtype": kdt, "parallelism": f"dp{world}",
                       "pipeline_depth": 1 if args.no_pipeline else args.pipeline},
            "roofline": {"bound": "hbm", "kernel": "attn_decode_kernel (decoder cross-attention, one layer, all utterances)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.workload),
                         "bytes_per_launch": nbytes.value, "us_per_launch": round(us.value, 2)},
            "decode_step": {"us": round(step_us.value, 1), "algorithmic_bytes": step_bytes.value,
                            "GBps": round(step_gbs, 1), "frac_of_hbm_peak": round(step_gbs / HBM_PEAK_GBS, 4)},
            "encoder": {"ms": round(enc_us.value / 1e3, 3), "TFLOPs": round(enc_flops / (enc_us.value * 1e-6) / 1e12, 1)},
        }
        if step4_us is not None:
            res["decode_step_4_in_flight"] = {"us_per_step_of_each_chain": round(step4_us, 1),
                                              "aggregate_GBps": round(4 * step_bytes.value / (step4_us * 1e-6) / 1e9, 1),
                                              "frac_of_hbm_peak": round(4 * step_bytes.value / (step4_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle) ...")
            res["cpu_baseline"] = cpu_baseline(cfg, weights, mel_host[0], DECODE_STEPS)
        print(json.dumps(res), flush=True)
    model.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
