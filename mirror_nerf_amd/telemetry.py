"""GPU telemetry for the measurement code (bench.py, scripts/): shader clock, socket power, hotspot temperature and
throttle state of one GPU, sampled every ~200 ms on a background thread while a timed region runs.

Why it exists: the dominant kernel runs at the package power limit, so the clock a box grants -- not the 2.4 GHz the MFMA
peak is quoted at -- sets its speed, and boxes of the pool differ by 14-17 % (profiles/DIARY.md 9.1).  A bench line has to say
whether a slow number is the box or the code: `summary()` gives min / median / max over the region, the OR of the
throttle bits and the fraction of the region spent at the power limit (ppt residency), for EACH timed leg.

Source, best first: librocm_smi64's gpu-metrics table read IN PROCESS through the ctypes bindings ROCm ships with
rocm-smi (/opt/rocm/libexec/rocm_smi/rsmiBindings.py -- struct layouts always match the installed library; ~50 us per
sample); else the hwmon / sysfs files of the card; else the `rocm-smi` command in a loop (~0.4 s per sample).  Every
failure is swallowed: telemetry never breaks a bench run, fields it could not read are None."""
import os
import re
import statistics
import subprocess
import sys
import threading
import time

_RSMI = None          # (lib, module) once loaded; False when unavailable


def _rsmi():
    global _RSMI
    if _RSMI is None:
        _RSMI = False
        try:
            root = os.environ.get("ROCM_PATH", "/opt/rocm")
            d = os.path.join(root, "libexec", "rocm_smi")
            if d not in sys.path:
                sys.path.insert(0, d)
            import rsmiBindings as B     # noqa: N812
            lib = B.initRsmiBindings(silent=True)
            if lib is not None and lib.rsmi_init(0) == 0:
                _RSMI = (lib, B)
        except BaseException:           # noqa: BLE001  (the bindings call exit() when the library is missing)
            _RSMI = False
    return _RSMI


def _rsmi_index(torch_index):
    """rocm-smi's device index of torch device `torch_index` (they differ under HIP_VISIBLE_DEVICES): matched by PCI address."""
    r = _rsmi()
    if not r:
        return None
    lib, B = r
    import ctypes
    n = ctypes.c_uint32(0)
    if lib.rsmi_num_monitor_devices(ctypes.byref(n)) != 0 or n.value == 0:
        return None
    if n.value == 1:
        return 0
    try:
        import torch
        p = torch.cuda.get_device_properties(torch_index)
        want = (int(p.pci_domain_id) << 32) | (int(p.pci_bus_id) << 8) | (int(p.pci_device_id) << 3)
        for i in range(n.value):
            bdf = ctypes.c_uint64(0)
            if lib.rsmi_dev_pci_id_get(i, ctypes.byref(bdf)) == 0 and (bdf.value & ~0x7 & 0xffffffff0000ffff) == want:
                return i
    except Exception:                   # noqa: BLE001
        pass
    return torch_index if torch_index < n.value else None


def _hwmon_dir(index):
    base = f"/sys/class/drm/card{index}/device/hwmon"
    try:
        for d in sorted(os.listdir(base)):
            return os.path.join(base, d)
    except OSError:
        return None
    return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class SmiSampler:
    """with SmiSampler(device_index) as s: <timed region>;  s.summary() -> dict."""

    def __init__(self, device_index=0, period_s=0.2):
        self.period = period_s
        self.dev = device_index
        self.samples = []            # dicts: t, sclk_mhz, power_w, temp_c, throttle, indep_throttle, ppt_acc, acc_counter
        self.source = None
        self.power_cap_w = None
        self._stop = threading.Event()
        self._thread = None

    # ---- one sample from each source
    def _sample_rsmi(self, idx):
        import ctypes
        lib, B = _rsmi()
        m = B.rsmi_gpu_metrics_t()
        if lib.rsmi_dev_gpu_metrics_info_get(idx, ctypes.byref(m)) != 0:
            return None
        na16 = 0xffff
        clks = [c for c in list(m.current_gfxclks) if c not in (0, na16)]
        sclk = (sum(clks) / len(clks)) if clks else (m.current_gfxclk if m.current_gfxclk not in (0, na16) else None)
        pw = m.current_socket_power if m.current_socket_power not in (0, na16) else \
            (m.average_socket_power if m.average_socket_power not in (0, na16) else None)
        temp = m.temperature_hotspot if m.temperature_hotspot not in (0, na16) else None
        big = 0xffffffffffffffff
        return {"sclk_mhz": sclk, "sclk_min_xcd_mhz": min(clks) if clks else None, "power_w": pw, "temp_c": temp,
                "throttle": None if m.throttle_status == 0xffffffff else int(m.throttle_status),
                "indep_throttle": None if m.indep_throttle_status == big else int(m.indep_throttle_status),
                "ppt_acc": None if m.ppt_residency_acc == big else int(m.ppt_residency_acc),
                "thm_acc": None if m.socket_thm_residency_acc == big else int(m.socket_thm_residency_acc),
                "acc_counter": None if m.accumulation_counter == big else int(m.accumulation_counter)}

    def _sample_sysfs(self, hw):
        f = _read_int(os.path.join(hw, "freq1_input"))
        p = _read_int(os.path.join(hw, "power1_input"))
        if p is None:
            p = _read_int(os.path.join(hw, "power1_average"))
        t = _read_int(os.path.join(hw, "temp2_input"))
        if t is None:
            t = _read_int(os.path.join(hw, "temp1_input"))
        if f is None and p is None:
            return None
        return {"sclk_mhz": None if f is None else f / 1e6, "power_w": None if p is None else p / 1e6,
                "temp_c": None if t is None else t / 1e3}

    def _sample_cli(self):
        try:
            txt = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks", "--showtemp"],
                                 capture_output=True, text=True, timeout=10).stdout
        except Exception:               # noqa: BLE001
            return None
        out = {}
        m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
        out["sclk_mhz"] = int(m.group(1)) if m else None
        m = re.search(r"(?:Current Socket|Average) Graphics Package Power \(W\):\s*([\d.]+)", txt)
        out["power_w"] = float(m.group(1)) if m else None
        m = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", txt)
        out["temp_c"] = float(m.group(1)) if m else None
        return out if (out["sclk_mhz"] is not None or out["power_w"] is not None) else None

    def _read_cap(self, idx):
        try:
            if idx is not None and _rsmi():
                import ctypes
                lib, _ = _rsmi()
                cap = ctypes.c_uint64(0)
                if lib.rsmi_dev_power_cap_get(idx, 0, ctypes.byref(cap)) == 0 and cap.value:
                    return cap.value / 1e6
            hw = _hwmon_dir(self.dev)
            if hw:
                c = _read_int(os.path.join(hw, "power1_cap"))
                if c:
                    return c / 1e6
        except Exception:               # noqa: BLE001
            pass
        return None

    def _run(self):
        try:
            idx = _rsmi_index(self.dev)
            hw = _hwmon_dir(self.dev)
            self.power_cap_w = self._read_cap(idx)
            if idx is not None and self._sample_rsmi(idx) is not None:
                self.source, take = "librocm_smi64 gpu_metrics, in process", (lambda: self._sample_rsmi(idx))
            elif hw is not None and self._sample_sysfs(hw) is not None:
                self.source, take = "hwmon sysfs", (lambda: self._sample_sysfs(hw))
            elif self._sample_cli() is not None:
                self.source, take = "rocm-smi command", self._sample_cli
            else:
                return
            while not self._stop.is_set():
                t = time.perf_counter()
                s = take()
                if s is not None:
                    s["t"] = t
                    self.samples.append(s)
                self._stop.wait(max(0.0, self.period - (time.perf_counter() - t)))
        except Exception:               # noqa: BLE001  (never break the run that is being measured)
            pass

    def __enter__(self):
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=15)
        return False

    # ---- reduction
    @staticmethod
    def _mmm(vals):
        vals = [v for v in vals if v is not None]
        if not vals:
            return None
        return {"min": round(min(vals), 1), "median": round(statistics.median(vals), 1), "max": round(max(vals), 1)}

    def summary(self):
        s = self.samples
        out = {"source": self.source, "period_s": self.period, "n_samples": len(s), "power_cap_w": self.power_cap_w,
               "sclk_mhz": self._mmm([x.get("sclk_mhz") for x in s]),
               "sclk_slowest_xcd_mhz": self._mmm([x.get("sclk_min_xcd_mhz") for x in s]),
               "power_w": self._mmm([x.get("power_w") for x in s]),
               "temp_hotspot_c": self._mmm([x.get("temp_c") for x in s])}
        thr = [x["throttle"] for x in s if x.get("throttle") is not None]
        ind = [x["indep_throttle"] for x in s if x.get("indep_throttle") is not None]
        out["throttle_status_or"] = None if not thr else hex(__import__("functools").reduce(lambda a, b: a | b, thr))
        out["indep_throttle_status_or"] = None if not ind else hex(__import__("functools").reduce(lambda a, b: a | b, ind))
        # residency accumulators tick with the firmware's accumulation counter: the share of the region's ticks during
        # which the power (ppt) / thermal limiter held the clock down
        out["ppt_residency_frac"] = out["thermal_residency_frac"] = None
        acc = [(x.get("acc_counter"), x.get("ppt_acc"), x.get("thm_acc")) for x in s if x.get("acc_counter") is not None]
        if len(acc) >= 2 and acc[-1][0] > acc[0][0]:
            d = acc[-1][0] - acc[0][0]
            if acc[0][1] is not None and acc[-1][1] is not None:
                out["ppt_residency_frac"] = round((acc[-1][1] - acc[0][1]) / d, 4)
            if acc[0][2] is not None and acc[-1][2] is not None:
                out["thermal_residency_frac"] = round((acc[-1][2] - acc[0][2]) / d, 4)
        return out

    def median_sclk(self):
        v = [x.get("sclk_mhz") for x in self.samples if x.get("sclk_mhz") is not None]
        return statistics.median(v) if v else None
