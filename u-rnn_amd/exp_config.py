"""Experiment YAMLs in the reference's own key names (config.py:55-213: --exp_config overrides the argparse defaults with the
YAML's top-level keys).  Only the keys the rollout / evaluation / SWP path reads are interpreted; unknown keys are kept."""
import yaml

# key: (type, default) -- defaults of the reference's argparse (config.py:70-160) for the keys this build reads
KEYS = {
    "data_root": (str, "../data/urbanflood24"), "test_list_file": (str, ""), "train_list_file": (str, ""),
    "location": (str, ""),      # config.py:121: one catchment ("" = every location under data_root), test.py:738
    "input_height": (int, 500), "input_width": (int, 500), "historical_nums": (int, 30),
    "flood_max": (float, 5000.0), "rain_max": (float, 6.0), "cumsum_rain_max": (float, 250.0), "flood_thres": (float, 150.0),
    "duration": (int, 360), "seq_num": (int, 12), "window_size": (int, 36), "batch_size": (int, 1), "cls_thred": (float, 0.5),
    "train_event": (bool, True), "all_seq_train": (bool, False), "full_window_size": (bool, False), "prewarming": (bool, False),
}


def load_exp_config(path, **overrides):
    """YAML file -> dict with every key of KEYS present and typed; CLI overrides (non-None) win, as in config.py:196-210."""
    with open(path) as fh:
        raw = yaml.safe_load(fh) or {}
    cfg = dict(raw)
    for k, (typ, default) in KEYS.items():
        v = raw.get(k, default)
        cfg[k] = typ(v) if v is not None else default
    for k, v in overrides.items():
        if v is not None:
            cfg[k] = v
    return cfg


def workload(cfg, spatial_rain=False):
    """(H, W, historical_nums, T, rain_max, cumsum_rain_max, spatial_rain): the tuple bench.py / the rollout engine is sized by."""
    return (cfg["input_height"], cfg["input_width"], cfg["historical_nums"], cfg["duration"], cfg["rain_max"], cfg["cumsum_rain_max"],
            bool(spatial_rain))
