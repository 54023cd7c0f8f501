#!/usr/bin/env python3
"""bench.py — throughput of the soundscope analyzer hot path on MI355X.

One step = one pass of the whole hot path (mid/side 4096-pt Hann FFT spectrum at hop 1024,
K-weighted gated loudness + LRA, 4x true peak, min-max decimation) over a batch of synthetic
48 kHz stereo f32 streams already resident in HBM, followed by the corpus gate (one all-reduce
of the 2x1000-bin histograms when N > 1).  Weak scaling: every rank holds `--streams` streams
(1024 x 10 s = BASELINE config 3 per GPU; 8 ranks = config 4's 8192 streams).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC: RCCL across processes needs this (inherited on the GPU boxes; set if missing)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(batch, rate, fft_n, hop, budget_s=15.0, all_cores_budget_s=8.0):
    """Time the CPU restatement (oracle, kind 'port') on a bounded sample of the same streams."""
    from oracle import pyoracle as po
    native = True
    try:
        po.build(native=True)
        po.lib(native=True)
    except Exception:
        native = False
    if native:      # keep whichever build is faster on this host (AVX-512 codegen can lose)
        x0 = batch.download_input(0)
        tt = []
        for nat in (False, True):
            t0 = time.perf_counter()
            po.analyze_stream(rate, x0, fft_n, hop, native=nat)
            tt.append(time.perf_counter() - t0)
        native = tt[1] < tt[0]
    n_streams = int(batch.cfg.n_streams)
    done, samples, t_used = 0, 0, 0.0
    while done < n_streams and t_used < budget_s:
        x = batch.download_input(done)
        t0 = time.perf_counter()
        po.analyze_stream(rate, x, fft_n, hop, want_fft=True, want_wave=True, native=native)
        t_used += time.perf_counter() - t0
        samples += x.size
        done += 1
    out = {"value": samples / t_used, "unit": "samples/s", "cores": 1, "kind": "port",
           "sample": f"{done} of the batch's streams ({samples} samples), single thread, "
                     f"gcc -O3{' -march=native' if native else ''}, full analyze_stream pass "
                     "(waveform + mid/side + 2 FFTs/window incl. the crate's stats sorts + meter with true peak)"}
    # the same pass on every host core, streams sharded over threads; ctypes drops the
    # GIL inside the C call.  Informational: the reference itself is single-threaded (main.rs:67).
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = len(os.sched_getaffinity(0))
        per = 2 if cores <= 64 else 1            # streams per core: the leg stays a few seconds even on a 256-core host
        xs = [batch.download_input(i % n_streams) for i in range(cores * per)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda x: po.analyze_stream(rate, x, fft_n, hop, want_fft=True, want_wave=True, native=native), xs))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(x.size for x in xs) / dt, "cores": cores, "streams": len(xs)}
    except Exception as e:            # never let the informational leg break the bench line
        out["all_cores"] = {"error": str(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU (weak scaling, the default)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong scaling instead: this many streams in total, sharded over the ranks "
                         "(BASELINE config 4: 8192; fits one GPU's 288 GB)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--fft-n", type=int, default=4096)
    ap.add_argument("--hop", type=int, default=1024)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import soundscope_amd as ssa
    from soundscope_amd import _lib as L
    from soundscope_amd.distributed import allreduce_histograms, corpus_gate, shard_streams

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (soundscope_amd has no CPU path)")
    # SS_BENCH_SHARED_GPU=1 + SS_BENCH_BACKEND=gloo: launch-path check on a 1-GPU box (all ranks on device 0,
    # collectives staged through host memory); the judged runs use one GPU per rank and RCCL
    backend = os.environ.get("SS_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("SS_BENCH_SHARED_GPU") == "1" else local_rank
    torch.cuda.set_device(dev)
    rc = L.lib().ss_set_device(dev)
    if rc:
        raise SystemExit("ss_set_device failed: " + L.lib().ss_last_device_error().decode())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            try:        # prove the communicator before the timed region; a broken RCCL setup must not cost the whole line
                probe = torch.ones(1, dtype=torch.int64, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                ok = int(probe.item()) == world
            except Exception as e:
                print(f"[bench] rank {rank}: RCCL all-reduce failed ({e!r})", file=sys.stderr, flush=True)
                ok = False
            if not ok:
                raise SystemExit("RCCL all-reduce over the node's GPUs failed; set SS_BENCH_BACKEND=gloo to stage the "
                                 "16 kB histogram exchange through host memory instead")
        else:
            dist.init_process_group(backend)
    host_staged = world > 1 and backend != "nccl"

    frames = int(round(args.seconds * args.rate))
    strong = args.total_streams > 0
    total_streams = args.total_streams if strong else args.streams * world
    first, count = shard_streams(total_streams, rank, world)
    b = ssa.Batch(args.rate, 2, count, frames, args.fft_n, args.hop, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, first)
    lay = b.layout
    hist = torch.zeros(2000, dtype=torch.int64, device="cuda")

    def step():
        b.run()
        b.histograms_to_device(hist.data_ptr())      # syncs the batch's stream
        if host_staged:
            h = hist.cpu()
            allreduce_histograms(h)
            hist.copy_(h)
        else:
            allreduce_histograms(hist)                 # RCCL over xGMI when world > 1
        if world > 1:
            torch.cuda.current_stream().synchronize()  # the next step refills `hist`: the collective must be done

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    b.timing_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_staged else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    b.sync()

    samples_per_step = total_streams * frames * 2
    value = samples_per_step * args.steps / dt
    corpus_i, corpus_lra = corpus_gate(hist.cpu().numpy())

    if rank == 0:
        # dominant kernel: the spectrum kernel.  Algorithmic bytes per launch:
        # every input f32 once + every retained bin once = 4 B/sample + 4*W*2*nbins per stream.
        fft_ms, fft_n_launch = b.timing_read(L.SS_KERNEL_FFT)
        alg_bytes = count * (frames * 2 * 4 + lay.n_windows * lay.fft_channels * lay.n_bins * 4)
        achieved = alg_bytes / (fft_ms / max(fft_n_launch, 1) * 1e-3) / 1e9 if fft_ms > 0 else None
        kernels = {}
        for k in range(L.SS_KERNEL_COUNT):
            ms, n = b.timing_read(k)
            kernels[L.lib().ss_batch_kernel_name(b._h, k).decode()] = round(ms / max(n, 1), 4)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "fft_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # second kernel of the pass: streaming + MFMA true peak.  HBM fraction live; pipe utilisation from the committed PMC run
        td_ms, td_n = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
        td = {"kernel": L.lib().ss_batch_kernel_name(b._h, L.SS_KERNEL_TIME_DOMAIN).decode(),
              "algorithmic_bytes_per_launch": count * frames * 2 * 4}
        if td_ms > 0:
            td["achieved_GBps"] = td["algorithmic_bytes_per_launch"] / (td_ms / max(td_n, 1) * 1e-3) / 1e9
            td["hbm_frac"] = td["achieved_GBps"] / HBM_PEAK_GBS
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "td_pmc.json")))["derived"]
            td.update({"mfma_busy_frac": d["mfma_busy_frac"], "valu_busy_frac": d["valu_busy_frac"],
                       "lds_busy_frac": d["lds_busy_frac"], "pmc_source": "profiles/td_pmc.json"})
        except Exception:
            pass
        out = {
            "metric": "audio samples/s analyzed (48 kHz stereo)", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32+f64",
            "data": "synthetic",
            "config": {"workload": f"{str(total_streams) + ' streams in total' if strong else str(args.streams) + ' streams/GPU'} x {args.seconds:g} s, {args.rate} Hz stereo f32 "
                                   f"(BASELINE config 3 per GPU; config 4 at 8 GPUs): mid/side {args.fft_n}-pt Hann FFT "
                                   f"hop {args.hop} + K-weighted gated LUFS/LRA + 4x true peak + min-max decimation "
                                   "+ corpus gate (1 all-reduce of 2x1000 u64)",
                       "streams_total": total_streams, "windows_per_stream": lay.n_windows, "bins": lay.n_bins,
                       "sharding": f"streams, {world} rank(s)", "collective": ("none" if world == 1 else ("rccl" if not host_staged else backend + " (host staged)")), "corpus_integrated_lufs": corpus_i,
                       "corpus_lra": corpus_lra, "kernel_ms": kernels, "time_domain_kernel": td},
            "roofline": {"bound": "hbm", "kernel": L.lib().ss_batch_kernel_name(b._h, L.SS_KERNEL_FFT).decode(),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        # PCIe-inclusive rate (informational, never `value`): host f32 -> HBM upload of a slice + its share of a pass
        try:
            k = min(count, 64)
            host = np.concatenate([b.download_input(i) for i in range(k)])
            t0 = time.perf_counter()
            b.upload(0, host)
            up = time.perf_counter() - t0
            out["config"]["pcie_inclusive_samples_per_s"] = host.size / (up + dt / args.steps * k / count)
            out["config"]["h2d_GBps"] = host.nbytes / up / 1e9
        except Exception:
            pass
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(b, args.rate, args.fft_n, args.hop)
            out["cpu_baseline"] = cb
            out["config"]["gpu_over_cpu_1core"] = value / cb["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
