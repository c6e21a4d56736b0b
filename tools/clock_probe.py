"""Shader clock while one GEMM kernel runs back to back (is the split-bf16 kernel power-throttled?):
python tools/clock_probe.py   - samples rocm-smi while each kernel loops for ~2.5 s."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from melissa_amd import _lib
lib = _lib.load()
M, N, K = 65536, 512, 512
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
Y = torch.empty(M, N, device="cuda")
scratch = torch.empty(6 * N * K + 256, dtype=torch.uint8, device="cuda")
def fp32(): lib.mel_gemm_f32(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, 0, _lib.current_stream_ptr())
first = [True]
def split():
    lib.mel_gemm_f32_split(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, 0, 2 if first[0] else 102, 0,
                           scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr())
    first[0] = False
def sample(out, stop):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            out.append([l.strip() for l in r.splitlines() if "sclk" in l or "Power" in l or "power" in l][:3])
        except Exception as e:       # noqa: BLE001
            out.append([repr(e)])
        time.sleep(0.3)
for name, fn in (("idle", None), ("exact fp32 64x64", fp32), ("split bf16 128x128", split)):
    out, stop = [], threading.Event()
    t = threading.Thread(target=sample, args=(out, stop)); t.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        if fn is None:
            time.sleep(0.1)
        else:
            for _ in range(50): fn()
            torch.cuda.synchronize(); n += 50
    stop.set(); t.join()
    dt = time.time() - t0
    rate = f"{n / dt:.0f} launches/s = {2.0 * M * N * K * n / dt / 1e12:.1f} TF" if n else ""
    print(name, rate)
    for s in out[1:5]: print("   ", s)
