"""Board power and shader clock while a kernel runs in a loop (is the forward power-limited?).  Polls the amdgpu hwmon / gpu_metrics files (falls back
to rocm-smi) from a thread while the main thread launches one workload back to back for a few seconds.
Usage: python tools/power_probe.py [seconds]"""
import glob
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def _read(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None


def find_sensors():
    s = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input"):
            if _read(os.path.join(hw, name)) is not None:
                s.setdefault("power_uw", os.path.join(hw, name))
        if _read(os.path.join(hw, "freq1_input")) is not None:
            s.setdefault("sclk_hz", os.path.join(hw, "freq1_input"))
        if _read(os.path.join(hw, "power1_cap")) is not None:
            s.setdefault("cap_uw", os.path.join(hw, "power1_cap"))
    return s


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
        return out.strip().replace("\n", " | ")[:400]
    except Exception as e:
        return f"rocm-smi failed: {e}"


def run(name, fn):
    sens = find_sensors()
    stop = [False]
    pw, ck, smi = [], [], []

    def poll():
        while not stop[0]:
            if "power_uw" in sens:
                v = _read(sens["power_uw"])
                if v and v.isdigit():
                    pw.append(int(v) / 1e6)
            if "sclk_hz" in sens:
                v = _read(sens["sclk_hz"])
                if v and v.isdigit():
                    ck.append(int(v) / 1e6)
            if not sens:
                smi.append(smi_sample())
            time.sleep(0.02 if sens else 0.5)

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    half = len(pw) // 2
    line = f"{name}: {ms:.4f} ms/launch over {n} launches"
    if pw:
        line += f" | power W: median(2nd half) {statistics.median(pw[half:]):.0f} max {max(pw):.0f} (n={len(pw)})"
    if ck:
        line += f" | sclk MHz: median(2nd half) {statistics.median(ck[len(ck) // 2:]):.0f} min {min(ck):.0f} max {max(ck):.0f}"
    cap = _read(sens.get("cap_uw", "")) if sens.get("cap_uw") else None
    if cap:
        line += f" | cap {int(cap) / 1e6:.0f} W"
    print(line, flush=True)
    for s in smi[-2:]:
        print("   ", s)
    return ms


def main():
    print("sensors:", find_sensors(), flush=True)
    print("idle:", smi_sample(), flush=True)
    torch.manual_seed(0)
    for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (1, 16384, 16, 128, False)):
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
        fl = 4 * B * H * S * S * D / (2 if causal else 1)
        ms = run(f"fwd B={B} S={S} causal={int(causal)}", lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None))
        print(f"    -> {fl / ms / 1e9:.0f} TFLOP/s")
        if causal:
            out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
            do = torch.randn_like(out)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ms = run("bwd same config", lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None))
            print(f"    -> {2.5 * fl / ms / 1e9:.0f} TFLOP/s")
    # a bandwidth-only workload for contrast
    x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
    run("copy 1 GiB", lambda: y.copy_(x))


if __name__ == "__main__":
    main()
