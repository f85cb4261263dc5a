"""Ray sets shared by the CPU and GPU parity tests."""
import numpy as np

import ntrace_amd as nt


def edge_rays(cam_extent=10.0):
    """Axis-parallel rays (zero direction components, +0 and -0), rays starting on geometry
    planes, degenerate rays (tmax < tmin), infinite / NaN tmax, tiny directions."""
    r = []

    def ray(o, d, tmin=0.0, tmax=1e30):
        r.append((o[0], o[1], o[2], tmin, d[0], d[1], d[2], tmax))
    g = np.linspace(-9.0, 9.0, 13)
    for x in g:
        for y in g:
            ray((x, y, -14.0), (0.0, 0.0, 1.0))
            ray((x, y, -14.0), (-0.0, 0.0, 1.0))
            ray((x, -14.0, y), (0.0, 1.0, -0.0))
            ray((-14.0, x, y), (1.0, 0.0, 0.0))
            ray((x, y, -15.0), (0.0, 0.0, 1.0))          # origin on the wall plane (e = 15)
            ray((x, y, 0.0), (0.0, 0.0, 1.0), 0.0, float("inf"))
            ray((x, y, 0.0), (1e-30, 1.0, 1e-38), 0.0, float("inf"))
            ray((x, y, 0.0), (0.3, 0.4, 0.5), 5.0, 4.0)  # degenerate
            ray((x, y, 0.0), (0.3, 0.4, 0.5), 0.0, float("nan"))
            ray((x, y, 0.0), (0.0, 0.0, 0.0))
            ray((x, y, 0.0), (0.6, 0.0, 0.8), 0.0, 3.0)  # short rays (AO-like)
    a = np.array(r, dtype=np.float32)
    return a.view(nt.RAY_DTYPE).reshape(-1)
