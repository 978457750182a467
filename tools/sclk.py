#!/usr/bin/env python
"""sclk.py -- the shader clock the GPU actually runs at while a region executes, sampled from the amdgpu hwmon node
(/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input, Hz) by a host thread.  bench.py brackets its timed regions with it: the
roofline's cycle counts (SQ_BUSY_CYCLES) are in shader cycles, the driver's clock is in milliseconds, and the chip moves between
~1.7 and 2.4 GHz with the load (profiles/r03/issue_cost_calibration_tables.txt: the MHz column), so converting one into the other
needs the clock OF THAT REGION, not a nominal one."""
import glob
import os
import threading
import time


def hwmon_freq_path(pci_bus_id=None):
    """freq1_input of the card with the given PCI bus id ('0000:05:00.0'); with one candidate (the GPU box) that one."""
    cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    if pci_bus_id:
        for c in cands:
            dev = os.path.realpath(os.path.join(os.path.dirname(c), "..", ".."))
            if os.path.basename(dev).lower() == pci_bus_id.lower():
                return c
    return cands[0] if cands else None


class SclkSampler:
    """with SclkSampler(path) as s: ...region...; s.mean_mhz, s.min_mhz, s.max_mhz, s.n"""

    def __init__(self, path, period_s=0.0005):
        self.path, self.period = path, period_s
        self.samples = []
        self._stop = threading.Event()
        self._t = None

    def _run(self):
        try:
            fd = os.open(self.path, os.O_RDONLY)
        except OSError:
            return
        try:
            while not self._stop.is_set():
                try:
                    os.lseek(fd, 0, os.SEEK_SET)
                    v = int(os.read(fd, 32).split()[0])
                    if v > 0:
                        self.samples.append(v)
                except (OSError, ValueError, IndexError):
                    pass
                time.sleep(self.period)
        finally:
            os.close(fd)

    def __enter__(self):
        if self.path:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t:
            self._t.join(timeout=1.0)
        return False

    @property
    def n(self):
        return len(self.samples)

    def stats(self):
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"mean_mhz": sum(s) / len(s) / 1e6, "min_mhz": s[0] / 1e6, "max_mhz": s[-1] / 1e6, "median_mhz": s[len(s) // 2] / 1e6, "samples": len(s),
                "source": self.path}


if __name__ == "__main__":
    p = hwmon_freq_path()
    print("hwmon node:", p)
    with SclkSampler(p) as s:
        time.sleep(1.0)
    print(s.stats())
