"""Fixed-grid solvers of torchdiffeq.odeint as called at reference flow_matching.py:62:
grid = the given t, one step per interval, returns the stacked trajectory."""
import torch


def odeint(func, y0, t, *, method="dopri5", **_):
    ys = [y0]
    y = y0
    for i in range(len(t) - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        if method == "euler":
            dy = dt * func(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            dy = dt * func(t0 + half, y + func(t0, y) * half)
        elif method == "rk4":
            k1 = func(t0, y)
            k2 = func(t0 + dt / 3, y + dt * k1 / 3)
            k3 = func(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = func(t1, y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        else:
            raise ValueError(method)
        y = y + dy
        ys.append(y)
    return torch.stack(ys)
