#!/usr/bin/env python
"""Development builds next to the product library, selected with URNN_LIB=<path>:
  tune   liburnn_hip_tune.so   -DURNN_TUNING (URNN_TUNE_* knobs live)
  trace  liburnn_hip_trace.so  -DURNN_TUNING -DURNN_TRACE (s_memtime stamps)
  NAME:-Dflag[,-Dflag...]      liburnn_hip_NAME.so with those flags (A/B builds)"""
import importlib.util, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(R, "u-rnn_amd", "build_ext.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
for arg in (sys.argv[1:] or ["tune", "trace"]):
    name, _, extra = arg.partition(":")
    flags = extra.split(",") if extra else ["-DURNN_TUNING"] + (["-DURNN_TRACE"] if name == "trace" else [])
    print(m.build(extra_flags=flags, out=os.path.join(R, "u-rnn_amd", f"liburnn_hip_{name}.so"), tag="_" + name))
