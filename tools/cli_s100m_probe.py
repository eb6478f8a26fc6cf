"""The command line on the S100M FASTQ (31.6 GB in /dev/shm): where do the seconds go, run after run?
python tools/cli_s100m_probe.py [reads]   -- on the GPU box.  Writes the file once, then runs kmerust with different pauses
between the runs, with KMERHIP_TRACE=1 (slow allocations and the host side of every count are printed)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import krust_amd
import bench

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
exe = os.path.join(ROOT, "krust_amd", "host", "kmerust")
stride, W = 151, 166 + 150
path = f"/dev/shm/kmerust_probe_{os.getpid()}.fq"
try:
    with open(path, "wb") as f:
        for r0 in range(0, reads, 10_000_000):
            nr = min(10_000_000, reads - r0)
            tb = torch.empty(nr * stride, dtype=torch.uint8, device=dev)
            tq = torch.empty(nr * stride, dtype=torch.uint8, device=dev)
            krust_amd.synth_reads_device(tb.data_ptr(), tq.data_ptr(), bench.SEED, bench.GENOME_LEN, 150, r0, nr, device=0,
                                         stream=torch.cuda.current_stream().cuda_stream)
            rec = torch.empty((nr, W), dtype=torch.uint8, device=dev)
            rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), 10
            r = torch.arange(r0, r0 + nr, device=dev)
            for j in range(9):
                rec[:, 10 - j] = ((r // 10 ** j) % 10 + 48).to(torch.uint8)
            rec[:, 12:162] = tb.view(nr, stride)[:, :150]
            rec[:, 162], rec[:, 163], rec[:, 164] = 10, ord("+"), 10
            rec[:, 165:315] = tq.view(nr, stride)[:, :150]
            rec[:, 315] = 10
            torch.cuda.synchronize()
            rec.cpu().numpy().tofile(f)
            del rec, tb, tq, r
    torch.cuda.empty_cache()
    print("file", os.path.getsize(path) / 1e9, "GB", flush=True)
    t0 = time.perf_counter(); subprocess.run(["cat", path], stdout=subprocess.DEVNULL); print("cat once:", round(time.perf_counter() - t0, 2), "s", flush=True)
    # PROBE_VARIANTS="A=1,B=2;C=3": environment settings to run one after the other (each with every pause), "" = the default
    variants = [dict(kv.split("=", 1) for kv in v.split(",") if kv) for v in os.environ.get("PROBE_VARIANTS", "").split(";")]
    for var in variants:
        env = dict(os.environ, KMERUST_TIMING="1", KMERHIP_TRACE="1", **var)
        print("variant", var or "default", flush=True)
        for pause in [float(x) for x in os.environ.get("PROBE_PAUSES", "3,3,10,10,0,0").split(",")]:
          time.sleep(pause)
          t0 = time.perf_counter()
          p = subprocess.run([exe, "21", path, "--format", "histogram", "-q"], capture_output=True, env=env, timeout=600)
          wall = time.perf_counter() - t0
          err = p.stderr.decode(errors="replace").splitlines()
          tj = [json.loads(l)["kmerust_timing"] for l in err if l.startswith('{"kmerust_timing"')]
          slow = [l for l in err if "took" in l or "host side" in l or "accumulation buffer" in l]
          print(f"pause {pause}: wall {wall:.2f} rc {p.returncode} chunk_kb {os.environ.get('KMERUST_TEXT_CHUNK_KB', 'default')}", {k: round(v, 3) for k, v in (tj[0] if tj else {}).items() if k.endswith("_s")}, flush=True)
          for l in slow[:12]:
              print("     ", l[:200], flush=True)
finally:
    if os.path.exists(path):
        os.remove(path)
