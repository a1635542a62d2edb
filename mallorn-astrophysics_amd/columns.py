"""Column-name tables of the feature frames (the consumer contract).

Every list reproduces the order in which the reference's per-object dict is filled, so a
DataFrame built from ``COLUMNS[set]`` has the same columns in the same order as the
reference extractor's output (checked against the live reference by
``tests/golden/make_golden.py`` and pinned in ``tests/golden/columns.json``).

The integer ids / bit masks below are the same constants as ``include/lcfe.h``.
"""
BANDS = ["u", "g", "r", "i", "z", "y"]

# feature-set ids (bit = 1 << id) -- keep in sync with include/lcfe.h
SET_STAT, SET_BAZIN, SET_POWERLAW, SET_TDE, SET_COLOR, SET_SHAPE, SET_PHYSICS, SET_GP2D, SET_GP1D, SET_RESEARCH = range(10)
SET_NAMES = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d", "gp1d", "research"]

_STAT17 = ["n_obs", "mean", "std", "min", "max", "median", "skew", "kurtosis", "amplitude", "mad",
           "iqr", "beyond_1std", "beyond_2std", "max_slope", "mean_snr", "time_span", "cadence_mean"]


def _stat():
    # statistical.py:168-222
    cols = [f"{p}_{s}" for p in BANDS + ["all"] for s in _STAT17]
    return cols + ["flux_ratio_g_r", "flux_ratio_r_i", "flux_ratio_i_z", "peak_band"]


_BAZIN8 = ["bazin_A", "bazin_t0", "bazin_tau_rise", "bazin_tau_fall", "bazin_B", "bazin_fit_chi2",
           "bazin_rise_fall_ratio", "bazin_peak_flux"]


def _bazin():
    # bazin_fitting.py:194-249
    cols = [f"{b}_{p}" for b in BANDS for p in _BAZIN8]
    return cols + ["bazin_rise_consistency", "bazin_fall_consistency", "bazin_avg_fit_chi2",
                   "bazin_fit_quality_dispersion"]


POWERLAW_MODELS = ["powerlaw_5_3", "powerlaw_1", "powerlaw_1_5", "powerlaw_2", "powerlaw_2_5",
                   "powerlaw_3", "powerlaw_0_5", "exponential", "linear"]


def _powerlaw():
    # train_v55_powerlaw.py:135-145,196-202
    return [f"{b}_{m}_r2" for b in "gri" for m in POWERLAW_MODELS]


def _tde():
    # tde_physics.py:355-374
    cols = []
    for a, b in (("g", "r"), ("r", "i")):
        cols += [f"{a}_{b}_color_var", f"{a}_{b}_color_range", f"{a}_{b}_color_trend"]
    for b in "gri":
        cols += [f"{b}_late_slope", f"{b}_late_flux_ratio", f"{b}_rebrightening"]
    for b in "gr":
        cols += [f"{b}_rise_shape", f"{b}_rise_rate"]
    cols += ["temp_stability", "temp_trend", "temp_late_vs_peak"]
    cols += ["r_decay_alpha", "r_decay_residual", "r_decay_alpha_late"]
    return cols


COLOR_PAIRS = [("g", "r"), ("r", "i"), ("u", "g"), ("i", "z")]
COLOR_EPOCHS = [("peak", 0), ("post_10d", 10), ("post_20d", 20), ("post_30d", 30), ("post_50d", 50),
                ("post_75d", 75), ("post_100d", 100), ("post_150d", 150), ("pre_10d", -10),
                ("pre_20d", -20)]


def _color():
    # colors.py:150-342
    cols = ["peak_mjd"]
    for e, _ in COLOR_EPOCHS:
        cols += [f"{a}_{b}_{e}" for a, b in COLOR_PAIRS]
    for a, b in COLOR_PAIRS:
        cols += [f"{a}_{b}_slope_50d", f"{a}_{b}_slope_100d"]
    for a, b in COLOR_PAIRS:
        cols += [f"{a}_{b}_std", f"{a}_{b}_range"]
    cols += [f"{b}_peak_flux" for b in BANDS]
    cols += [f"{a}_{b}_peak_flux_ratio" for a, b in COLOR_PAIRS]
    cols += ["g_r_peak_lag", "r_i_peak_lag"]
    cols += ["g_r_curvature", "r_i_curvature"]
    for a, b in (("g", "r"), ("r", "i")):
        cols += [f"{a}_{b}_late_stability", f"{a}_{b}_late_mean"]
    cols += ["temp_peak", "temp_post_30d", "temp_post_75d", "temp_post_150d"]
    cols += ["temp_slope_early", "temp_slope_mid", "temp_slope_late", "temp_stability"]
    return cols


def _shape():
    # lightcurve_shape.py:204-330
    cols = []
    for b in BANDS:
        cols += [f"{b}_rise_time", f"{b}_fade_time_50", f"{b}_fade_time_25", f"{b}_asymmetry",
                 f"{b}_duration_50", f"{b}_duration_25", f"{b}_power_law_alpha",
                 f"{b}_power_law_residual"]
    cols += ["peak_time_spread", "peak_time_std", "optical_mean_rise_time", "optical_mean_fade_time",
             "optical_mean_power_alpha", "rise_time_consistency", "fade_time_consistency",
             "all_rise_time", "all_fade_time_50", "all_asymmetry", "all_power_law_alpha",
             "all_power_law_residual", "flux_p10", "flux_p25", "flux_p75", "flux_p90",
             "flux_concentration"]
    return cols


def _physics():
    # physics_based.py:316-456
    cols = ["stetson_j_gr", "stetson_j_ri", "stetson_j_gi", "stetson_k_g", "stetson_k_r",
            "stetson_k_i"]
    cols += [f"r_sf_tau_{t}" for t in (1, 5, 10, 30, 100)] + ["r_sf_slope"]
    for b in "gri":
        cols += [f"{b}_rest_duration", f"{b}_rest_rise", f"{b}_rest_fade"]
    cols += ["temp_at_peak", "temp_post_50d", "temp_evolution"]
    cols += ["r_bazin_amplitude", "r_bazin_t0", "r_bazin_rise_approx", "r_bazin_fall_approx",
             "r_bazin_plateau"]
    cols += ["mean_snr", "median_snr", "excess_variance"]
    return cols


def _gp2d():
    # multiband_gp.py:182-188,224-277
    cols = ["gp2d_amplitude", "gp2d_time_scale", "gp2d_wave_scale", "gp2d_log_likelihood",
            "gp2d_time_wave_ratio"]
    for e in (0, 20, 50, 100):
        cols += [f"gp_flux_g_{e}d", f"gp_flux_r_{e}d", f"gp_flux_i_{e}d", f"gp_gr_color_{e}d",
                 f"gp_ri_color_{e}d"]
    return cols + ["gp_gr_slope_50d", "gp_gr_slope_100d"]


def _gp1d():
    # gaussian_process.py:189-246 (per-band scikit-learn GP; bands g, r, i, z)
    cols = [f"{b}_{k}" for b in "griz" for k in ("gp_length_scale", "gp_amplitude", "gp_noise", "gp_log_likelihood")]
    return cols + ["gp_ls_ratio_gr", "gp_ls_ratio_ri", "gp_mean_length_scale", "gp_std_length_scale",
                   "gp_mean_amplitude"]


def _research():
    # research_features.py:57-64,150-158,175-180,265-269,371-375,472-478 (the v115 "research" features)
    pl = ["powerlaw_alpha", "powerlaw_alpha_deviation_53", "powerlaw_alpha_deviation_512", "powerlaw_chi2",
          "powerlaw_residual_std", "powerlaw_fit_success"]
    cols = [f"{b}_{k}" for b in "gri" for k in pl]
    cols += ["optical_mean_powerlaw_alpha", "optical_std_powerlaw_alpha", "optical_mean_deviation_53"]
    cols += ["nuclear_smoothness", "nuclear_concentration", "nuclear_variability_ratio", "nuclear_position_score"]
    cols += ["g_r_color_at_peak", "g_r_color_peak_to_late", "r_i_color_at_peak", "r_i_color_peak_to_late"]
    cols += ["mhps_10d", "mhps_30d", "mhps_100d", "mhps_10_100_ratio", "mhps_30_100_ratio", "mhps_dominant_scale"]
    cols += ["luminosity_distance_mpc", "peak_luminosity", "luminosity_amplitude", "mean_luminosity",
             "luminosity_decline_rate"]
    return cols


COLUMNS = {"stat": _stat(), "bazin": _bazin(), "powerlaw": _powerlaw(), "tde": _tde(),
           "color": _color(), "shape": _shape(), "physics": _physics(), "gp2d": _gp2d(), "gp1d": _gp1d(),
           "research": _research()}
NCOLS = {k: len(v) for k, v in COLUMNS.items()}
assert NCOLS == {"stat": 123, "bazin": 52, "powerlaw": 27, "tde": 25, "color": 83, "shape": 65,
                 "physics": 32, "gp2d": 27, "gp1d": 21, "research": 40}, NCOLS

# integer-valued columns of the statistics frame (int64 in the reference's DataFrame)
STAT_INT_COLUMNS = [f"{p}_n_obs" for p in BANDS + ["all"]] + ["peak_band"]
