"""Sweep scripts describe a variant as LPG_<NAME> environment variables (the round-1 library re-read them on every
launch).  The library now has an option table instead (include/lpg_hip.h: lpg_set_option) and the release build never
looks at the environment: sync() pushes the current LPG_* variables into the table (unset = default).  Measurement-only
variants (spmm_mode = 2) exist in the -DLPG_TUNING build only: use_tuning_build() builds and selects it -- call it
BEFORE importing lpgnn_amd."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SKIP = {"LPG_LIB_VARIANT", "LPG_CHECK_INDICES", "LPG_BENCH_BACKEND", "LPG_BENCH_FORCE_DIST", "LPG_TWOWL_SMALL",
         "LPG_TWOWL_PARALLEL", "LPG_TEST_EDGES"}


def use_tuning_build():
    os.environ["LPG_LIB_VARIANT"] = "tuning"
    subprocess.run([sys.executable, os.path.join(ROOT, "link-prediction-gnn_amd", "csrc", "build.py"), "--tuning"], check=True)


def sync():
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from lpgnn_amd import _lib
    _lib.check(_lib.load().lpg_reset_options(), "reset_options")
    for k, v in os.environ.items():
        if k.startswith("LPG_") and k not in _SKIP and v != "":
            try:
                _lib.set_option(k[4:].lower(), int(v))
            except (_lib.LpgError, ValueError) as e:
                print(f"[profiles/_opts] {k}={v} ignored: {e}", file=sys.stderr)
