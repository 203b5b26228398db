"""Independent numerical reference for ProDMP trajectories: the DMP ordinary
differential equation of the ProDMP paper (Li et al., RA-L 2023, cited at
/root/reference/README.md:221-233), integrated with scipy -- NOT the closed
form, the pre-computed tables or any function of ``oracle/prodmp_oracle.py`` /
``tce_rl_amd/mp/prodmp.py``.

Per dof, with z = tau * dy/dt:

    tau^2 y'' = alpha (beta (g - y) - tau y') + f(x),     beta = alpha / 4
    f(x)      = x * sum_b phi_b(x) w_b / sum_b phi_b(x)
    x(t)      = exp(-alpha_phase * max(t - delay, 0) / tau)   (canonical system)
    phi_b(x)  = exp(-h_b (x - c_b)^2 / 2)

In scaled time s = max(t - delay, 0) / tau this is
    y'' = alpha^2/4 (g - y) - alpha y' + f(x(s)),      ' = d/ds.

Conventions of the basis generator that a paper does not fix (they come from
the constructor surface at mprl/util/util_mp.py:22-46 and are stated here as
the specification under test): centres equally spaced in time on
[delay, delay + tau] (``num_basis_outside`` = 0) and mapped to phase space,
c_b = x(t_b); widths h_b = basis_bandwidth_factor / (c_{b+1} - c_b)^2 (last one
repeated); ``auto_scale_basis``: every weight (and the goal) is multiplied by
1 / max_s |unit response| over s in [0, pre_compute_length_factor] sampled at
the pre-compute grid, the unit response being the solution from rest
(y = y' = 0) for w_b = 1 (resp. g = 1) alone; then ``weights_scale`` /
``goal_scale``; ``relative_goal``: g_abs = g_scaled + y0.
"""
import numpy as np
from scipy.integrate import solve_ivp


class DMPODE:
    def __init__(self, num_basis, tau, alpha_phase, alpha, dt,
                 basis_bandwidth_factor, delay=0.0, weights_scale=1.0,
                 goal_scale=1.0, auto_scale_basis=True, relative_goal=False,
                 pre_compute_length_factor=5, num_dof=None,
                 num_basis_outside=0, rtol=1e-11, atol=1e-13):
        assert num_basis_outside == 0
        self.nb, self.tau, self.delay = int(num_basis), float(tau), float(delay)
        self.ax, self.a = float(alpha_phase), float(alpha)
        self.relative_goal = relative_goal
        self.rtol, self.atol = rtol, atol
        t_c = self.delay + np.linspace(0.0, self.tau, self.nb)
        self.c = np.exp(-self.ax * (t_c - self.delay) / self.tau)
        gap = np.diff(self.c)
        gap = np.concatenate([gap, gap[-1:]]) if self.nb > 1 else np.ones(1)
        self.h = basis_bandwidth_factor / gap ** 2
        self.scale = np.ones(self.nb + 1)
        if auto_scale_basis:
            n = int(pre_compute_length_factor) * int(round(self.tau / dt)) + 1
            grid = np.linspace(0.0, float(pre_compute_length_factor), n)
            for b in range(self.nb + 1):
                w = np.zeros(self.nb)
                g = 0.0
                if b < self.nb:
                    w[b] = 1.0
                else:
                    g = 1.0
                y = self._solve(grid, 0.0, 0.0, 0.0, w, g)[0]
                self.scale[b] = 1.0 / np.abs(y).max()
        self.scale[:-1] *= weights_scale
        self.scale[-1] *= goal_scale

    def forcing(self, s, w):
        x = np.exp(-self.ax * s)
        phi = np.exp(-0.5 * self.h * (x - self.c) ** 2)
        if self.nb > 1:
            phi = phi / phi.sum()
        return x * float(phi @ w)

    def _solve(self, s_eval, s0, y0, z0, w, g):
        """y(s), y'(s) at the increasing scaled times s_eval >= s0."""
        a = self.a

        def rhs(s, u):
            return [u[1], 0.25 * a * a * (g - u[0]) - a * u[1]
                    + self.forcing(s, w)]
        s_eval = np.asarray(s_eval, dtype=float)
        hi = max(float(s_eval[-1]), s0 + 1e-9)
        # query times before the phase delay all map to s = 0: solve on the
        # distinct values
        uniq, inv = np.unique(np.clip(s_eval, s0, hi), return_inverse=True)
        sol = solve_ivp(rhs, (s0, hi), [y0, z0], method="DOP853",
                        t_eval=uniq, rtol=self.rtol, atol=self.atol)
        assert sol.success, sol.message
        return sol.y[0][inv], sol.y[1][inv]

    def trajectory(self, times, params, init_time, init_pos, init_vel):
        """One env: times [T] (increasing), params [dof * (nb + 1)] laid out
        [w_1..w_nb, g] per dof, init_pos / init_vel [dof] -> pos, vel [T, dof]."""
        times = np.asarray(times, dtype=float)
        dof = len(init_pos)
        th = np.asarray(params, dtype=float).reshape(dof, self.nb + 1)
        s = np.clip((times - self.delay) / self.tau, 0.0, None)
        s0 = max((float(init_time) - self.delay) / self.tau, 0.0)
        pos = np.empty((len(times), dof))
        vel = np.empty_like(pos)
        for d in range(dof):
            w = th[d, :-1] * self.scale[:-1]
            g = th[d, -1] * self.scale[-1]
            if self.relative_goal:
                g = g + float(init_pos[d])
            y, z = self._solve(s, s0, float(init_pos[d]),
                               float(init_vel[d]) * self.tau, w, g)
            pos[:, d], vel[:, d] = y, z / self.tau
        return pos, vel
