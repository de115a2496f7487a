"""bench.py's clock / power telemetry (mirror_nerf_amd/telemetry.py) must never break a run: on a host without a GPU every
source fails quietly and the summary holds None fields; the reduction of a sample series (min / median / max, OR of the
throttle words, residency fractions from the firmware's accumulators) is checked on synthetic samples."""
import time

from mirror_nerf_amd.telemetry import SmiSampler


def test_sampler_is_silent_without_a_gpu():
    with SmiSampler(0, 0.05) as s:
        time.sleep(0.15)
    out = s.summary()
    assert set(out) >= {"source", "n_samples", "sclk_mhz", "sclk_slowest_xcd_mhz", "power_w", "temp_hotspot_c",
                        "throttle_status_or", "ppt_residency_frac", "thermal_residency_frac"}
    if out["n_samples"] == 0:
        assert out["sclk_mhz"] is None and out["power_w"] is None and s.median_sclk() is None


def test_summary_of_a_synthetic_series():
    s = SmiSampler(0, 0.2)
    s.source = "synthetic"
    for i, (clk, slow, pw, thr) in enumerate([(2100, 2000, 1300, 0x0), (2150, 2040, 1310, 0x4), (2200, 2080, 1290, 0x1)]):
        s.samples.append({"t": 0.2 * i, "sclk_mhz": clk, "sclk_min_xcd_mhz": slow, "power_w": pw, "temp_c": 60 + i,
                          "throttle": thr, "indep_throttle": None, "ppt_acc": 1000 + 80 * i, "thm_acc": 50,
                          "acc_counter": 5000 + 100 * i})
    out = s.summary()
    assert out["sclk_mhz"] == {"min": 2100, "median": 2150, "max": 2200}
    assert out["sclk_slowest_xcd_mhz"]["median"] == 2040
    assert out["power_w"]["max"] == 1310 and out["temp_hotspot_c"]["min"] == 60
    assert out["throttle_status_or"] == "0x5" and out["indep_throttle_status_or"] is None
    assert out["ppt_residency_frac"] == 0.8 and out["thermal_residency_frac"] == 0.0
    assert s.median_sclk() == 2150
