"""What distinguishes one GPU box from another at the moment of a measurement (bench.py puts it into its JSON line): clocks,
power, temperature, partition modes and memory, read from sysfs (amdgpu hwmon) -- no privileges, no child process.  Every
field is optional: what cannot be read is left out."""
import glob
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _current_level(text):
    """pp_dpm_* lists levels as `1: 2400Mhz *`; the starred one is current"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return float(line.split(":", 1)[1].replace("*", "").strip().lower().replace("mhz", ""))
            except (ValueError, IndexError):
                return None
    return None


def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(dev, "vendor")) == "0x1002" and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            out.append(dev)
    return out


def pick(pci=None, index=0):
    """the sysfs device directory of a card: by PCI address ("0000:05:00.0", what the process's HIP device reports -- a box may
    show all cards of its host while the process owns one), else the index-th amdgpu card"""
    devs = cards()
    if pci:
        want = pci.lower()
        for dev in devs:
            if os.path.basename(os.path.realpath(dev)).lower() == want:
                return dev
    return devs[index] if index < len(devs) else None


def snapshot(index=0, pci=None):
    """one card's state now: a flat dict of numbers and short strings"""
    dev = pick(pci, index)
    if dev is None:
        return {}
    snap = {"card": os.path.basename(os.path.dirname(dev)), "pci": os.path.basename(os.path.realpath(dev)), "matched_by_pci": bool(pci) and os.path.basename(os.path.realpath(dev)).lower() == pci.lower()}
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk"), ("socclk_mhz", "pp_dpm_socclk")):
        v = _current_level(_read(os.path.join(dev, name)))
        if v is not None:
            snap[key] = v
    for key, name in (("busy_percent", "gpu_busy_percent"), ("mem_busy_percent", "mem_busy_percent"), ("vram_used_bytes", "mem_info_vram_used"),
                      ("vram_total_bytes", "mem_info_vram_total"), ("compute_partition", "current_compute_partition"),
                      ("memory_partition", "current_memory_partition"), ("perf_level", "power_dpm_force_performance_level")):
        v = _read(os.path.join(dev, name))
        if v is not None:
            snap[key] = int(v) if v.isdigit() else v
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for key, name, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("power_cap_w", "power1_cap", 1e-6),
                                 ("temp_edge_c", "temp1_input", 1e-3), ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3),
                                 ("hwmon_sclk_mhz", "freq1_input", 1e-6), ("hwmon_mclk_mhz", "freq2_input", 1e-6)):
            v = _read(os.path.join(hw, name))
            if v is not None and key not in snap:
                try:
                    snap[key] = round(float(v) * scale, 2)
                except ValueError:
                    pass
    return snap


class Sampler:
    """Samples a card's clocks / power / temperature on a thread of its own while a measurement runs (every `period` seconds;
    a handful of small sysfs reads each): what the card did DURING the timed steps, not before or after."""

    FIELDS = ("sclk_mhz", "hwmon_sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_junction_c", "temp_mem_c", "busy_percent", "mem_busy_percent")

    def __init__(self, index=0, pci=None, period=0.02):
        import threading
        self.index, self.pci, self.period = index, pci, period
        self.rows, self._stop, self._thread = [], threading.Event(), None

    def __enter__(self):
        import threading
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            snap = snapshot(self.index, self.pci)
            self.rows.append({k: snap[k] for k in self.FIELDS if isinstance(snap.get(k), (int, float))})
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self):
        out = {"samples": len(self.rows), "period_s": self.period}
        for k in self.FIELDS:
            vals = [r[k] for r in self.rows if k in r]
            if vals:
                out[k] = {"min": min(vals), "mean": round(sum(vals) / len(vals), 2), "max": max(vals)}
        return out


if __name__ == "__main__":
    import json
    print(json.dumps({"cards": cards(), "snapshot": snapshot()}, indent=1))
