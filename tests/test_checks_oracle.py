"""Oracle of the post-step checks against the fixtures captured from the reference classes
(oracle/capture_golden_checks.py): SafetyChecker / SafetyMonitor sequences and the solution-quality gate."""
import os

import numpy as np
import pytest

from oracle import checks_np as CK

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfgs(d, name):
    if name == "default":
        return CK.CheckerConfig(), CK.MonitorConfig()
    c, m = d["custom_checker_limits"], d["custom_monitor_limits"]
    return (CK.CheckerConfig((c[0], c[1]), (c[2], c[3]), c[4], c[5], c[6]),
            CK.MonitorConfig((m[0], m[1]), (m[2], m[3]), m[4], (m[5], m[6]), (m[7], m[8])))


@pytest.mark.parametrize("name", ["default", "custom"])
def test_safety_sequences_match_reference(name):
    d = np.load(os.path.join(GOLD, "checks_safety_seq.npz"))
    v, f, ld = d["voltages"], d["frequency"], d["loadings"]
    ccfg, mcfg = _cfgs(d, name)
    cs, ms = CK.CheckerState(), CK.MonitorState()
    dt = float(d[f"{name}_dt"])
    g = lambda k: d[f"{name}_{k}"]
    seen = set()
    for t in range(v.shape[0]):
        c = CK.checker_step(ccfg, cs, v[t], f[t], ld[t], dt)
        np.testing.assert_array_equal(c["voltage_low"], g("c_mask_low")[t].astype(bool))
        np.testing.assert_array_equal(c["voltage_high"], g("c_mask_high")[t].astype(bool))
        np.testing.assert_array_equal(c["line_overload"], g("c_mask_overload")[t].astype(bool))
        for mine, ref in (("n_voltage_low", "c_voltage_low"), ("n_voltage_high", "c_voltage_high"), ("frequency_low", "c_freq_low"),
                          ("frequency_high", "c_freq_high"), ("n_line_overload", "c_line_overload"), ("voltage_rate_violation", "c_voltage_rate"),
                          ("frequency_rate_violation", "c_freq_rate"), ("total", "c_total"), ("is_safe", "c_is_safe"), ("severity", "c_severity")):
            np.testing.assert_array_equal(np.asarray(c[mine]).astype(np.int64), g(ref)[t], err_msg=f"{mine} t={t}")
        vr = g("c_voltage_rate")[t].astype(bool); fr = g("c_freq_rate")[t].astype(bool)
        np.testing.assert_array_equal(c["voltage_rate"][vr], g("c_voltage_rate_value")[t][vr])      # same float operations: exact
        np.testing.assert_array_equal(c["frequency_rate"][fr], g("c_freq_rate_value")[t][fr])
        m = CK.monitor_step(mcfg, ms, v[t], f[t], ld[t])
        np.testing.assert_array_equal(m["voltage_high"], g("m_mask_high")[t].astype(bool))
        np.testing.assert_array_equal(m["voltage_low"], g("m_mask_low")[t].astype(bool))
        np.testing.assert_array_equal(m["line_overload"], g("m_mask_overload")[t].astype(bool))
        for mine, ref in (("n_voltage_emergency", "m_emergency_count"), ("frequency_high", "m_freq_high"), ("frequency_low", "m_freq_low"),
                          ("frequency_emergency", "m_freq_emergency"), ("total_violations", "m_total"),
                          ("emergency_action_required", "m_action_required"), ("consecutive_violations", "m_consecutive"),
                          ("emergency_mode", "m_emergency_mode")):
            np.testing.assert_array_equal(np.asarray(m[mine]).astype(np.int64), g(ref)[t], err_msg=f"{mine} t={t}")
        seen |= set(c["severity"].tolist())
    assert {0, 1, 2, 3} <= seen                               # every severity the fixture can reach was exercised
    assert g("m_emergency_mode").any() and not g("m_emergency_mode").all()
    assert g("c_voltage_rate").any() and g("c_freq_rate").any()


def test_quality_gate_matches_reference():
    d = np.load(os.path.join(GOLD, "checks_quality.npz"))
    q = CK.quality(d["converged"], d["iterations"], d["max_mismatch"], d["bus_voltages"], d["line_loadings"], d["line_flows"],
                   float(d["tolerance"]))
    np.testing.assert_array_equal(q, d["quality"])
    assert len(set(np.round(d["quality"], 6).tolist())) >= 8      # the multipliers were all exercised
