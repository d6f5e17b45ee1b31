#!/usr/bin/env python
"""bench.py — realtime factor (xRT) of the four-pass speech-mastering path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run, one rank per
GPU).  A step = all four passes over one 60-min 48 kHz mono f32 file that is already resident in HBM
(BASELINE.json configs[1]); files shard one per GPU, no data-path collective (scaling: weak).  Rank 0 prints ONE
JSON line with `roofline` (dominant kernel by measured time) and `cpu_baseline` (oracle port, bounded sample, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

# seven HIP streams per context (main, four analysis chains, two early-start streams): a hardware queue each (read at HIP runtime
# init); more queues than that oversubscribe the queue slots once several contexts share a GPU (jt_api.cpp, jt_open)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # before torch initialises HIP (jt_open asks for the same; see the note there)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(sample_seconds, sr):
    """Time the CPU oracle (single-threaded C restatement of the FFmpeg filters — a 'port', NOT the reference
    binary: Go + FFmpeg 8.1 are not available here) on a bounded sample of the same synthetic workload."""
    import numpy as np
    from jivetalking_amd import synth
    from oracle import orc
    orc.lib()
    x = synth.speech_like(sample_seconds, sr, seed=1234)
    t0 = time.perf_counter()
    xd = x.astype(np.float64)
    # Pass 1
    orc.astats(xd, sr); orc.aspectralstats(x, sr); e1 = orc.ebur128(xd, sr, True, True)
    # Pass 2
    y = orc.biquad_f32(orc.biquad_f32(x, 0, 80.0, sr), 1, 20500.0, sr)
    y = orc.anlmdn(y, sr)
    y = orc.afftdn(y, sr, 12.0, -60.0)
    yd = orc.acompressor(orc.agate(y.astype(np.float64), sr), sr)
    yf = yd.astype(np.float32)
    orc.astats(yd, sr); orc.aspectralstats(yf, sr); e2 = orc.ebur128(yf.astype(np.float64), sr, True, True)
    s16 = orc.f64_to_s16(orc.swr_f64(yf.astype(np.float64), sr, 44100, True))
    # Pass 3
    up = orc.swr_f32(s16.astype(np.float32) / 32768.0, 44100, 192000, True)
    m = orc.loudnorm_measure(up.astype(np.float64), 192000, True)
    # Pass 4
    g = 10 ** ((-16.0 - m["input_i"]) / 20.0)
    z = orc.alimiter(orc.adeclick(s16.astype(np.float64) / 32768.0 * g, 44100), 44100, 10 ** (-1.9 / 20), 1.0, 50.0)
    zf = z.astype(np.float32)
    orc.astats(z, 44100); orc.aspectralstats(zf, 44100); orc.ebur128(zf.astype(np.float64), 44100, True, True)
    orc.f64_to_s16(zf.astype(np.float64))
    dt = time.perf_counter() - t0
    return {"value": round(sample_seconds / dt, 2), "unit": "xRT", "cores": 1, "kind": "port",
            "sample": f"{sample_seconds:.0f} s of the same synthetic 48 kHz mono speech, all four passes, scalar C oracle",
            "reference_self_published": "~18 xRT/file (README.md:105-125, unknown CPU, 3 files in flight)"}


def limiter_prefix_leg(eng, x_dev, n, sr, seconds, base, hostlogic):
    """The same step on a file whose peaks would pass the ceiling after the loudnorm gain, so that the plan needs the alimiter prefix
    (normalise.go:452-497): Pass 3 then measures the limited f64 signal and Pass 4 starts from it.  Real speech usually takes this path;
    the bench's synthetic talker (crest factor ~12 dB) does not.  40 plosive-like bursts a minute are added to the bench signal.
    Reported beside `value`, never part of it."""
    import numpy as np
    import torch
    dev = x_dev.device
    y = x_dev.clone()
    g = torch.Generator(device=dev).manual_seed(7)
    pos = torch.randint(sr, n - sr, (max(1, int(seconds / 60.0 * 40)),), device=dev, generator=g)
    t = torch.arange(960, device=dev)
    burst = (0.35 * torch.hann_window(960, device=dev) * torch.sin(2 * np.pi * 180.0 * t / sr)).float()
    idx = (pos[:, None] + t[None, :]).reshape(-1)
    y.index_add_(0, idx, burst.repeat(pos.numel()))
    torch.cuda.synchronize()
    eng.attach_device_pcm(y.data_ptr(), n, sr, 1, keepalive=y)
    ts = []
    for it in range(5):
        t0 = time.perf_counter(); r = hostlogic.process_audio(eng, base, 4096); ts.append(time.perf_counter() - t0)
    tm = eng.timers()
    eng.attach_device_pcm(x_dev.data_ptr(), n, sr, 1, keepalive=x_dev)
    return {"ms_per_step": round(min(ts[2:]) * 1e3, 2), "xRT": round(seconds / min(ts[2:]), 1), "limiter_needed": int(r.limiter.needed),
            "pass_ms": {"pass1": round(tm["pass1_ms"], 2), "pass2": round(tm["pass2_ms"], 2), "pass3": round(tm["pass3_ms"], 2), "pass4": round(tm["pass4_ms"], 2)},
            "output_lufs": round(r.output_lufs, 2), "output_dbtp": round(r.output_tp_db, 2)}


def e2e_legs(eng, x_dev, n, sr, seconds, base, hostlogic, Engine, device):
    """End-to-end legs, reported next to `value` and never part of it (`value` has the input resident in HBM):
      pcie : pinned host f32 file -> H2D -> four passes -> D2H of the s16 output into pinned memory, per file
      file : jt_process_file on a 16-bit FLAC in /dev/shm -> "<name>-LUFS-16-processed.flac" beside it (read, GPU decode, four
             passes, GPU encode, write), with and without the STREAMINFO MD5 (one dependent chain on one host core), and a batch
             of files through jt_process_files (MD5 on), where each worker's MD5 hides behind the other workers' GPU phases."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    out = {}
    xh = x_dev.cpu().pin_memory()
    xn = xh.numpy()
    m_cap = int(-(-n * 147 // 160)) + 16
    yh = torch.empty(m_cap, dtype=torch.int16).pin_memory(); yn = yh.numpy()
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        eng.upload_pcm(xn, sr, 1)
        r = hostlogic.process_audio(eng, base, 4096)
        got = eng.download_s16_into(4, yn)
        ts.append(time.perf_counter() - t0)
    ts = ts[1:]
    out["pcie"] = {"ms_per_file": round(min(ts) * 1e3, 2), "xRT": round(seconds / min(ts), 1), "h2d_bytes": int(n * 4), "d2h_bytes": int(got * 2),
                   "note": "pinned host buffers both ways; upload, four passes, download of the final s16, per file, nothing overlapped"}
    d = tempfile.mkdtemp(prefix="jtbench", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        pcm = (x_dev * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
        src = os.path.join(d, "episode.flac")
        open(src, "wb").write(eng.op_flac_encode(pcm, sr, md5=True))
        res = {}
        for md5 in (False, True):
            tt = []
            for it in range(3):
                t0 = time.perf_counter()
                r, outp, io = hostlogic.process_file(eng, src, base, 4096, md5=md5)
                tt.append(time.perf_counter() - t0)
            res["md5" if md5 else "no_md5"] = {"ms_per_file": round(min(tt[1:]) * 1e3, 2), "xRT": round(seconds / min(tt[1:]), 1),
                                               "io_ms": dict(zip(["read", "decode", "encode", "write"], [round(v, 2) for v in io]))}
        nb = 6
        paths = []
        for k in range(nb):
            pk = os.path.join(d, f"batch{k}.flac"); shutil.copyfile(src, pk); paths.append(pk)
        t0 = time.perf_counter()
        failed, fr = hostlogic.process_files(paths, device=device, in_flight=3, base=base, md5=True)
        tb = time.perf_counter() - t0
        res["batch_md5"] = {"files": nb, "in_flight": 3, "failed": int(failed), "ms_per_file": round(tb / nb * 1e3, 2), "xRT": round(nb * seconds / tb, 1),
                            "note": "jt_process_files, first-file allocations of the three worker handles included"}
        res["input"] = {"format": "FLAC 16-bit mono", "bytes": os.path.getsize(src), "location": d.split("/jtbench")[0]}
        out["file"] = res
    finally:
        shutil.rmtree(d, ignore_errors=True)
    eng.attach_device_pcm(x_dev.data_ptr(), n, sr, 1, keepalive=x_dev)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--minutes", type=float, default=60.0, help="audio minutes per file (BASELINE configs[1] = 60)")
    ap.add_argument("--cpu-sample", type=float, default=20.0, help="seconds of audio for the CPU oracle baseline (0 = skip)")
    ap.add_argument("--rate", type=int, default=48000, help="input sample rate (BASELINE configs[4]: 96000)")
    ap.add_argument("--channels", type=int, default=1, help="input channels, 1 or 2 (configs[4]: 2, down-mixed on the device)")
    ap.add_argument("--e2e", type=int, default=1, help="also time the end-to-end legs outside `value` (PCIe-inclusive, file to file); 0 = skip")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="extra measurement (not `value`): K files per GPU processed concurrently, one context + host thread each "
                         "(BASELINE configs[3], throughput saturation); reported as `saturation`")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from jivetalking_amd import Engine, synth, hostlogic, shard
    sr = args.rate
    seconds = args.minutes * 60.0
    x = synth.speech_like_torch(seconds, sr, seed=1000 + rank, device=f"cuda:{local_rank}")
    n = x.numel()
    if args.channels == 2:
        # second channel: the same talker 0.15 ms later and 2 dB down (interleaved L R L R ...)
        x = torch.stack([x, 0.8 * torch.roll(x, 14)], dim=1).contiguous().view(-1)
    torch.cuda.synchronize()
    eng = Engine(local_rank)
    eng.attach_device_pcm(x.data_ptr(), n, sr, args.channels, keepalive=x)

    base = hostlogic.default_config()

    def step():
        # ProcessAudio mirror in C++ (jt_process_audio): four GPU passes + VAD / AdaptConfig / limiter planning between them
        return hostlogic.process_audio(eng, base, 4096)

    res = None
    for _ in range(args.warmup):
        res = step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nlm_ms, dk_ms, p_ms = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        t = eng.timers()
        nlm_ms.append(t["nlm_ms"]); dk_ms.append(t["declick_ms"]); p_ms.append([t["pass1_ms"], t["pass2_ms"], t["pass3_ms"], t["pass4_ms"]])
    barrier()
    dt = time.perf_counter() - t0
    dt_rank = dt
    dt = shard.max_over_ranks(dt, device=f"cuda:{local_rank}")
    per_rank_ms = shard.gather_over_ranks(dt_rank / args.steps * 1e3, device=f"cuda:{local_rank}")

    sat = None
    if args.in_flight > 1:
        # K independent contexts (own streams, own buffers) share the GPU; the C calls release the GIL
        import threading
        engs = [eng] + [Engine(local_rank) for _ in range(args.in_flight - 1)]
        for e2 in engs[1:]:
            e2.attach_device_pcm(x.data_ptr(), n, sr, 1, keepalive=x)
            hostlogic.process_audio(e2, base, 4096)                       # warm-up (allocations)
        def worker(e2, k):
            for _ in range(k):
                hostlogic.process_audio(e2, base, 4096)
        barrier()
        t1 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(e2, args.steps)) for e2 in engs]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        barrier()
        dt2 = shard.max_over_ranks(time.perf_counter() - t1, device=f"cuda:{local_rank}")
        sat = {"files_in_flight_per_gpu": args.in_flight, "xRT": round(world * args.in_flight * args.steps * seconds / dt2, 1),
               "ms_per_file": round(dt2 / (args.in_flight * args.steps) * 1e3, 3)}
        for e2 in engs[1:]: e2.close()
    if rank == 0:
        import numpy as np
        m = int(-(-n * 147 // 160))
        alg_bytes_file = 8 * n + 8 * m                       # 4 unavoidable sweeps
        value = world * args.steps * seconds / dt
        nlm_avg_s = float(np.mean(nlm_ms)) / 1e3
        K, S = 288, 96
        nlm_bytes = 8 * n                                    # anlmdn: read f32 + write f32 per sample
        nlm_flops = n * (2 * S) * 6                          # patch-distance recurrence: 2 sub, 2 mul, 2 add per (sample, offset)
        pm = np.mean(np.array(p_ms), axis=0)
        nlm_roof = {"kernel": "k_anlmdn_pair3", "bound": "hbm", "achieved": round(nlm_bytes / nlm_avg_s / 1e9, 2),
                    "peak": 8000, "unit": "GB/s", "frac": round(nlm_bytes / nlm_avg_s / 1e9 / 8000, 5), "traffic": None,
                    "note": "vector-FP32 bound, not HBM bound (SURVEY §8d): add/mul-only recurrence (FFmpeg's unfused f32 order), "
                            "so the applicable peak is the non-FMA packed rate.  This launch is the early Pass-2 head (jt_pass2_prefetch_after_pass1): "
                            "it is queued behind the Pass-1 analysis kernels and shares the GPU with their tails",
                    "valu": {"achieved_TFLOPs": round(nlm_flops / nlm_avg_s / 1e12, 2), "peak_TFLOPs_no_fma": 78.6,
                             "frac": round(nlm_flops / nlm_avg_s / 1e12 / 78.6, 4)},
                    "avg_launch_ms": round(nlm_avg_s * 1e3, 3)}
        dk_avg_s = float(np.mean(dk_ms)) / 1e3
        if dk_avg_s > nlm_avg_s:
            dk_bytes = 16 * m                                  # adeclick: read f64 + write f64 per 44.1 kHz sample
            roof = {"kernel": "k_adeclick", "bound": "hbm", "achieved": round(dk_bytes / dk_avg_s / 1e9, 2), "peak": 8000,
                    "unit": "GB/s", "frac": round(dk_bytes / dk_avg_s / 1e9 / 8000, 5), "traffic": None,
                    "note": "dominant kernel; not bandwidth bound: a window is an AR fit, a detector and a banded LDL^T solve whose pivots "
                            "are a dependent chain (k_adeclick_fast: matrix-pipe autocorrelation, register-blocked detector, "
                            "diagonal-major LDS ring; ~35 k wave-instructions per window, SIMDs ~25 % busy, the rest is LDS / "
                            "dependent-issue latency at 10 waves per CU: profiles/r02_pmc_issue.txt, DESIGN.md s4/s9)",
                    "avg_launch_ms": round(dk_avg_s * 1e3, 3), "repaired_samples": int(eng.timers()["declick_repaired"]),
                    "heavy_windows": int(eng.timers()["declick_heavy_windows"])}
        else:
            roof, nlm_roof = nlm_roof, None
        # HBM traffic per launch from the committed PMC passes (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs of this same command); FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note
        try:
            pmc_file = "r02_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) else "r01_pmc_traffic.json"
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["kernels"]
            for r_ in (roof, nlm_roof):
                # a kernel launched as several template instances per step (adeclick's three capacity levels) is summed
                ks = [v for k, v in pmc.items() if k == r_["kernel"] or k.startswith(r_["kernel"] + "<") or k.startswith(r_["kernel"] + "_fast<")] if r_ else []
                if ks:
                    r_["traffic"] = int(sum(2 * k_["FETCH_SIZE_KB_max_call"] + k_["WRITE_SIZE_KB_max_call"] for k_ in ks) * 1024)
                    r_["traffic_source"] = "profiles/" + pmc_file + " (PMC pass of this command, not collected live)"
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "realtime factor (xRT) on 48 kHz mono speech, 1/2/4/8 GPUs; LUFS error vs ref",
            "value": round(value, 1), "unit": "xRT", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
            "config": {"workload": f"1 x {args.minutes:g} min {sr / 1000:g} kHz {'stereo' if args.channels == 2 else 'mono'} f32 PCM per GPU, full 4-pass pipeline, input resident in HBM",
                       "files_per_gpu_per_step": 1, "adeclick": "on (t=1.7 w=55 o=50 m=s, the reference default)",
                       "adaptive": "full host mirror (VAD, speech election, AdaptConfig, band RMS) in C++",
                       "pass2_spec": res.pass2_spec.decode()},
            "result": {"output_lufs": round(res.output_lufs, 3), "output_dbtp": round(res.output_tp_db, 3),
                       "input_lufs": round(res.input_lufs, 3), "within_target": bool(res.within_target)},
            "pass_ms": {"pass1": round(float(pm[0]), 3), "pass2": round(float(pm[1]), 3),
                        "pass3": round(float(pm[2]), 3), "pass4": round(float(pm[3]), 3)},
            "stage_wall_ms": dict(zip(["pass1", "intervals_vad", "bands", "adapt", "pass2", "regions2", "plan", "pass3", "pass4", "regions4"],
                                      [round(float(v), 3) for v in res.stage_ms])),
            "pipeline_hbm": {"algorithmic_bytes_per_file": alg_bytes_file,
                             "achieved_GBps": round(alg_bytes_file * world * args.steps / dt / 1e9, 2), "peak_GBps": 8000},
            "roofline": roof,
            "second_kernel": nlm_roof,
        }
        if sat is not None:
            out["saturation"] = sat
        if world == 1:
            # the file formats either side of the path, outside the timed region: Pass-4 output -> .flac image on
            # the GPU (with and without the host-side STREAMINFO MD5), and that image decoded again on the GPU by a second handle
            enc = [eng.flac_encode(4, md5=False, return_info=True)[1] for _ in range(3)]
            image, enc_md5 = eng.flac_encode(4, md5=True, return_info=True)
            e3 = Engine(local_rank)
            dec = [e3.load_audio(image) for _ in range(3)]
            e3.close()
            out["io_legs"] = {
                "flac_encode": {"gpu_ms": round(min(i["gpu_ms"] for i in enc), 3), "total_ms": round(min(i["total_ms"] for i in enc), 3),
                                "bytes": enc_md5["bytes"], "ratio_vs_s16": round(enc_md5["bytes"] / (2.0 * enc_md5["total_samples"]), 4),
                                "frames": enc_md5["frames"], "md5_host_ms": round(enc_md5["md5_ms"], 1),
                                "note": "jt_flac_encode(stage 4): analyse + scan + emit kernels, D2H of the image into pinned memory; "
                                        "the MD5 is one dependent chain on one host core and is optional (JT_FLAC_MD5)"},
                "flac_decode": {"gpu_ms": round(min(d["gpu_ms"] for d in dec), 3), "total_ms": round(min(d["total_ms"] for d in dec), 3),
                                "frames": dec[0]["flac_frames"], "candidates": dec[0]["flac_candidates"],
                                "note": "jt_load_audio of that image (44.1 kHz mono s16): H2D from pageable memory, find + parse + "
                                        "decode + finish kernels, host chain walk"},
            }
        if world == 1 and args.e2e and args.channels == 1:
            out["e2e"] = e2e_legs(eng, x, n, sr, seconds, base, hostlogic, Engine, local_rank)
            out["limiter_prefix"] = limiter_prefix_leg(eng, x, n, sr, seconds, base, hostlogic)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, sr)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
