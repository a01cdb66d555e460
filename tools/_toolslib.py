"""Point audiopure_amd at the -DAP_TOOLS build (timing-only ap_debug_* hooks); build it if missing.  Import first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

os.environ["AUDIOPURE_HIP_LIB"] = G.build_hip(tools=True)
