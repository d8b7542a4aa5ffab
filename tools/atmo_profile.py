"""The atmosphere precompute (pt_atmosphere_render -> atmo_precompute_kernel) on its own: N renders of a size^2 x 6 cube, timed with the
library's timer.  Run under rocprofv3 (--kernel-trace --stats, and a separate --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES pass) by
tools/round_profiles.sh; prints one JSON line.   python tools/atmo_profile.py <size> [renders]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pt = pkg.PathTracer(None, 64, 64, 8, 1, 20.0, 0.14)
atmo = pkg.AtmosphericScatterer(size, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)  # (renders once)
pt.Synchronize()
pt.TimerBegin()
for _ in range(n):
    atmo.Render()
ms = pt.TimerEnd() / n
texels = 6 * size * size
print(json.dumps({"size": size, "renders": n, "ms_per_render": round(ms, 4), "texels": texels, "ns_per_texel": round(ms * 1e6 / texels, 3),
                  "inner_iterations_per_s": round(texels * 50 * 15 / (ms * 1e-3) / 1e9, 2), "csrc_hash": pkg.native.csrc_hash()}))
pt.Dispose()
