"""What box did this run on?  (the same kernel on the same workload took 4.5 ms on one box of the pool
and 5.4 ms on another, and nothing in the line said which box it was.)

`snapshot()`  -- once, before the timed region: rocm-smi's clocks / power cap / partition modes and the sysfs files behind
                 them (whichever of the two the container exposes).
`Sampler`     -- a thread that reads the live shader / memory clock and the socket power from sysfs (hwmon freq1/freq2/
                 power1) every 20 ms WHILE the timed region runs; a read is a few microseconds of host time, nothing
                 touches the GPU's queues.
Both return plain dicts for bench_detail.json; bench.py puts a few numbers of them into the line."""
import glob
import json
import os
import subprocess
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _amd_cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(dev, "vendor")) == "0x1002":
            out.append(dev)
    return out


def hip_device_sysfs(device_index=0):
    """sysfs directory of HIP device `device_index`: a box of the pool has 8 cards and shows the container ONE of them,
    so `card0` is usually somebody else's GPU (seen in round 5: an idle 95 MHz next to a kernel at full tilt).  The
    PCI bus id of the HIP device names the right directory."""
    bdf = None
    try:
        import ctypes
        hip = None
        # the HIP runtime this process already runs on (torch ships its own copy: never load a second one beside it)
        loaded = []
        try:
            with open("/proc/self/maps") as f:
                loaded = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
        except OSError:
            pass
        for name in loaded + ["libamdhip64.so", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"]:
            try:
                hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if hip is not None:
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
                bdf = buf.value.decode().lower()
    except Exception:
        bdf = None
    if bdf:
        path = os.path.join("/sys/bus/pci/devices", bdf)
        if os.path.isdir(path):
            return path, bdf
    return None, bdf


def _device_dir(device_index=0):
    path, _ = hip_device_sysfs(device_index)
    if path:
        return path
    cards = _amd_cards()
    return cards[min(device_index, len(cards) - 1)] if cards else None


def _hwmon(dev):
    h = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
    return h[0] if h else None


def _active_level(txt):
    """'0: 132Mhz\n1: 2400Mhz *' -> ('2400Mhz', [levels])"""
    if not txt:
        return None, None
    levels = [l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l]
    active = [l.rstrip("* ").strip() for l in levels if l.endswith("*")]
    return (active[0] if active else None), [l.rstrip("* ").strip() for l in levels]


def sysfs_state(card_index=0):
    cards = _amd_cards()
    dev = _device_dir(card_index)
    if not dev:
        return {"available": False}
    hw = _hwmon(dev)
    sclk, sclk_levels = _active_level(_read(os.path.join(dev, "pp_dpm_sclk")))
    mclk, mclk_levels = _active_level(_read(os.path.join(dev, "pp_dpm_mclk")))
    fclk, _ = _active_level(_read(os.path.join(dev, "pp_dpm_fclk")))
    st = {"available": True, "device": dev, "pci_bus_id": hip_device_sysfs(card_index)[1], "cards": len(cards),
          "sclk_active": sclk, "sclk_levels": sclk_levels, "mclk_active": mclk, "mclk_levels": mclk_levels,
          "fclk_active": fclk,
          "perf_level": _read(os.path.join(dev, "power_dpm_force_performance_level")),
          "compute_partition": _read(os.path.join(dev, "current_compute_partition")),
          "memory_partition": _read(os.path.join(dev, "current_memory_partition")),
          "vram_total_bytes": _read(os.path.join(dev, "mem_info_vram_total")),
          "vram_used_bytes": _read(os.path.join(dev, "mem_info_vram_used")),
          "vbios": _read(os.path.join(dev, "vbios_version"))}
    if hw:
        def uw(name):
            v = _read(os.path.join(hw, name))
            return float(v) / 1e6 if v and v.lstrip("-").isdigit() else None
        st["power_cap_W"] = uw("power1_cap")
        st["power_cap_max_W"] = uw("power1_cap_max")
        st["power_W"] = uw("power1_average") or uw("power1_input")
        for k, name in (("sclk_MHz", "freq1_input"), ("mclk_MHz", "freq2_input")):
            v = _read(os.path.join(hw, name))
            st[k] = float(v) / 1e6 if v and v.isdigit() else None
        t = _read(os.path.join(hw, "temp1_input"))
        st["temp_C"] = float(t) / 1e3 if t and t.lstrip("-").isdigit() else None
    thp = _read("/sys/kernel/mm/transparent_hugepage/enabled")
    st["host_thp"] = thp
    return st


def rocm_smi_state(timeout=25):
    """rocm-smi's own view (one subprocess; a second of host time, before anything is timed)."""
    exe = None
    for c in ("/opt/rocm/bin/rocm-smi", "rocm-smi"):
        if c.startswith("/") and not os.path.exists(c):
            continue
        exe = c
        break
    if not exe:
        return {"available": False}
    cmd = [exe, "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showcomputepartition",
           "--showmemorypartition", "--showmeminfo", "vram", "--json"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"available": False, "error": repr(e)}
    txt = r.stdout.strip()
    try:
        data = json.loads(txt[txt.index("{"):])
        card = data.get("card0") or next(iter(data.values()))
        return {"available": True, "card0": card, "cards": len([k for k in data if k.startswith("card")])}
    except (ValueError, StopIteration):
        return {"available": bool(txt), "raw": txt[-1500:], "stderr": r.stderr[-300:], "rc": r.returncode}


def firmware_state(timeout=25):
    """driver + firmware versions (MEC / RLC / SMC ...): cards of the pool differ by 20 % on one kernel at equal clocks, power
    cap and copy ceiling (profiles/r05_headline_ab.txt addendum 3) -- the versions are what is left to compare"""
    exe = "/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else "rocm-smi"
    try:
        r = subprocess.run([exe, "--showfwinfo", "--showdriverversion", "--json"], capture_output=True, text=True, timeout=timeout)
        txt = r.stdout.strip()
        data = json.loads(txt[txt.index("{"):])
        card = data.get("card0") or next(iter(data.values()))
        out = {"available": True, "card0": card}
        if "system" in data:
            out["system"] = data["system"]
        return out
    except Exception as e:   # noqa: BLE001  (a box without rocm-smi, an output that is not JSON: say so, do not fail a bench)
        return {"available": False, "error": repr(e)[:200]}


def snapshot(card_index=0):
    return {"sysfs": sysfs_state(card_index), "rocm_smi": rocm_smi_state(), "firmware": firmware_state()}


class Sampler:
    """with Sampler() as s: <timed region>;  s.summary() -> {sclk_MHz: {min, median, max}, ...}"""

    def __init__(self, card_index=0, period_s=0.02):
        dev = _device_dir(card_index)
        self.hw = _hwmon(dev) if dev else None
        self.period = period_s
        self.rows = []
        self._stop = threading.Event()
        self._th = None

    def _loop(self):
        files = {"sclk_MHz": ("freq1_input", 1e6), "mclk_MHz": ("freq2_input", 1e6),
                 "power_W": ("power1_average", 1e6), "power_in_W": ("power1_input", 1e6), "temp_C": ("temp1_input", 1e3)}
        while not self._stop.is_set():
            row = {}
            for k, (name, div) in files.items():
                v = _read(os.path.join(self.hw, name))
                if v and v.lstrip("-").isdigit():
                    row[k] = float(v) / div
            if row:
                self.rows.append(row)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.hw:
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join(timeout=1.0)
        return False

    def summary(self):
        if not self.rows:
            return {"available": False}
        out = {"available": True, "samples": len(self.rows)}
        for k in ("sclk_MHz", "mclk_MHz", "power_W", "power_in_W", "temp_C"):
            v = sorted(r[k] for r in self.rows if k in r)
            if v:
                out[k] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        return out


if __name__ == "__main__":
    with Sampler() as s:
        time.sleep(0.2)
    print(json.dumps({"snapshot": snapshot(), "sampled": s.summary()}, indent=1))
