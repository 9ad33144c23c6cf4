#!/usr/bin/env python3
"""Power and clock telemetry of the GPU while the bench kernel runs (is the trace kernel power-limited?).

    python tools/power_clock.py [name=lib.so ...]        (default: the in-tree library)

For every library: a few seconds of back-to-back `Raytracer.trace(10 M rays)` on the bench scene while a sampler thread
reads, every 50 ms, whatever the box offers an ordinary user:
  * sysfs: /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,freq1_input,temp*_input},
    pp_dpm_sclk (current level), gpu_busy_percent;
  * else `amd-smi metric --power --clock --json` / `rocm-smi --showpower --showclocks --json`.
Prints per arm: launches, mean kernel ms, and min / mean / max of every quantity found during the loaded window, plus the
idle reading before it.  If nothing is readable it says so (then the counter-derived clock of tools/clock_probe.sh stays
the only evidence).
"""
import glob
import json
import os
import pathlib
import shutil
import subprocess
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent


def sysfs_sources():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "vendor")):
            continue
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        files = {}
        for pat, key, scale in (("hwmon/hwmon*/power1_average", "power_W", 1e-6), ("hwmon/hwmon*/power1_input", "power_W", 1e-6),
                                ("hwmon/hwmon*/freq1_input", "sclk_MHz", 1e-6), ("hwmon/hwmon*/temp1_input", "temp_C", 1e-3),
                                ("gpu_busy_percent", "busy_pct", 1.0)):
            for f in glob.glob(os.path.join(dev, pat)):
                try:
                    float(open(f).read().strip())
                    files.setdefault(key, (f, scale))
                except (OSError, ValueError):
                    pass
        dpm = os.path.join(dev, "pp_dpm_sclk")
        if os.access(dpm, os.R_OK):
            files["dpm_sclk"] = (dpm, None)
        if files:
            out.append((dev, files))
    return out


def read_sysfs(files):
    r = {}
    for key, (f, scale) in files.items():
        try:
            txt = open(f).read()
        except OSError:
            continue
        if scale is None:  # pp_dpm_sclk: "1: 2100Mhz *"
            for line in txt.splitlines():
                if line.strip().endswith("*"):
                    try:
                        r["dpm_sclk_MHz"] = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                    except (IndexError, ValueError):
                        pass
        else:
            try:
                r[key] = float(txt.strip()) * scale
            except ValueError:
                pass
    return r


def read_smi():
    for cmd in (["amd-smi", "metric", "--power", "--clock", "--json"], ["rocm-smi", "--showpower", "--showclocks", "--json"]):
        exe = shutil.which(cmd[0]) or (f"/opt/rocm/bin/{cmd[0]}" if os.path.exists(f"/opt/rocm/bin/{cmd[0]}") else None)
        if not exe:
            continue
        try:
            o = subprocess.run([exe] + cmd[1:], capture_output=True, text=True, timeout=5)
            if o.returncode == 0 and o.stdout.strip():
                return cmd[0], o.stdout
        except Exception:
            pass
    return None, None


def flatten(obj, prefix=""):
    out = {}
    if isinstance(obj, dict):
        for k, v in obj.items():
            out.update(flatten(v, f"{prefix}{k}."))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            out.update(flatten(v, f"{prefix}{i}."))
    else:
        try:
            out[prefix[:-1]] = float(str(obj).split()[0])
        except (ValueError, IndexError):
            pass
    return out


def main():
    arms = [a.split("=", 1) for a in sys.argv[1:]] or [["in-tree", "optrace_amd/csrc/liboptrace_hip.so"]]
    srcs = sysfs_sources()
    smi_name, smi_txt = (None, None) if srcs else read_smi()
    print(f"telemetry sources: sysfs {[d for d, _ in srcs]} keys {[sorted(f) for _, f in srcs]}; smi: {smi_name}")
    if not srcs and not smi_name:
        print("NO power / clock telemetry is readable on this box (no amdgpu hwmon files, no amd-smi / rocm-smi): "
              "the counter-derived clock (tools/clock_probe.sh) is the only evidence.")

    for name, lib in arms:
        env = dict(os.environ, OPTRACE_AMD_LIB=str((ROOT / lib).resolve()))
        child = subprocess.Popen([sys.executable, "-c", CHILD, str(ROOT)], env=env, stdout=subprocess.PIPE, text=True)
        samples, idle, stop = [], [], threading.Event()
        phase = {"v": "start"}

        def sampler():
            while not stop.is_set():
                t = time.time()
                if srcs:
                    r = {}
                    for k, (_, files) in enumerate(srcs):
                        for kk, v in read_sysfs(files).items():
                            r[f"gpu{k}.{kk}"] = v
                else:
                    nm, txt = read_smi()
                    try:
                        r = flatten(json.loads(txt)) if txt else {}
                        r = {k: v for k, v in r.items() if any(s in k.lower() for s in ("power", "clk", "clock", "sclk", "freq"))}
                    except Exception:
                        r = {}
                (samples if phase["v"] == "load" else idle).append((t, r))
                time.sleep(0.05)

        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        result, mine = None, None
        for line in child.stdout:
            line = line.strip()
            if line.startswith("PCI "):  # the device the child runs on -> its index among the sysfs sources
                for k, (d, _) in enumerate(srcs):
                    if os.path.realpath(d).lower().endswith(line[4:].lower()):
                        mine = f"gpu{k}."
            elif line == "LOAD_BEGIN":
                phase["v"] = "load"
            elif line == "LOAD_END":
                phase["v"] = "done"
            elif line.startswith("{"):
                result = json.loads(line)
        child.wait()
        stop.set()
        th.join()
        print(f"=== {name}: {result}   (this process's device: {mine or 'unknown -- all devices of the host listed'})")
        for label, ss in (("idle", idle[:10]), ("loaded", samples[len(samples) // 4:])):  # skip the ramp of the loaded window
            keys = sorted({k for _, r in ss for k in r if mine is None or k.startswith(mine)})
            for k in keys:
                v = [r[k] for _, r in ss if k in r]
                if v:
                    print(f"  {label:7s} {k:28s} n={len(v):3d}  min {min(v):9.2f}  mean {sum(v)/len(v):9.2f}  max {max(v):9.2f}")


CHILD = r'''
import sys, time, json, ctypes as C
root = sys.argv[1]
sys.path[:0] = [root, root + "/tests"]
import torch
import optrace_amd as ot
from optrace_amd import _capi
import scenes
lib = _capi.load_library()
pr = torch.cuda.get_device_properties(0)
try:
    print("PCI %04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id), flush=True)
except AttributeError:
    pass
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot, seed=None)
    RT.trace(100_000)
    _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))
    torch.cuda.synchronize()
    time.sleep(1.0)
    print("LOAD_BEGIN", flush=True)
    ms, ks, t0 = C.c_double(), [], time.time()
    while time.time() - t0 < 4.0:
        RT.trace(10_000_000)
        _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle, C.byref(ms)))
        ks.append(ms.value)
    print("LOAD_END", flush=True)
k = ks[len(ks) // 2:]
print(json.dumps({"launches": len(ks), "kernel_ms_mean_second_half": sum(k) / len(k), "kernel_ms_first": ks[0]}), flush=True)
'''

if __name__ == "__main__":
    main()
