"""CPU restatement of the ProDMP trajectory generator.  No reference vectors
exist (third-party arithmetic, absent): pinned against an independent scipy
integration of the DMP ODE instead (tests/prodmp_ode.py, test_prodmp_ode_cpu.py).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The arithmetic is not in ``/root/reference``: it lives in the un-vendored
dependency ``mp_pytorch==0.1.4`` (``conda_env.sh:41``).  This file restates the
published algorithm (Li et al., "ProDMP: A Unified Perspective on Dynamic and
Probabilistic Movement Primitives", RA-L 2023; cited at ``README.md:221-233``)
behind the reference's call-site surface:

* constructor surface               ``mprl/util/util_mp.py:11-46``
* ``sample_trajectories`` / ``get_traj_pos`` / ``get_traj_vel``
                                    ``mprl/rl/policy/temporal_correlated_policy.py:74-92``
* ``update_inputs`` / ``get_traj_pos(flat_shape=True)`` / ``get_traj_pos_cov``
                                    ``mprl/rl/policy/temporal_correlated_policy.py:188-192``

Model (per dof, in scaled time s = (t - delay)/tau):
    y'' + alpha y' + alpha^2/4 y = alpha^2/4 g + x(s) phi(s)^T w
    x(s) = exp(-alpha_phase * max(s, 0)),  phi = normalised RBFs in phase space
Closed form  y(s) = c1 y1 + c2 y2 + Phi(s)^T [w; g],  y1 = exp(-alpha s/2),
y2 = s y1, Phi_w = y2 p2 - y1 p1 (p = cumulative trapezoid integrals of the
forcing bases on a grid of step dt/tau over 5 tau, linearly interpolated at
query times), Phi_g = y2 q2 - y1 q1 (closed form).  c1, c2 are fixed by
(y, dy/dt) at ``init_time`` which makes the trajectory affine in theta=[w; g]:
    pos(t) = xi1(t) y0 + xi2(t) tau v0 + H(t) theta
Choices that cannot be checked against the absent package (documented in
DESIGN.md): left-bounded (not upper-clipped) linear phase; covariance
regulariser 1e-4 on the diagonal; ``relative_goal`` means the absolute goal is
g_abs = scale_g * theta_g + y0 (y0 rides on the *unscaled* goal basis, which
tends to 1); auto-scale = 1 / max|basis| per column.
"""
import torch


def torch_lerp_rows(table, idx):
    """Linear interpolation of rows of ``table`` [M, ...] at float indices,
    same clipping and the same two-branch lerp formula torch.lerp uses
    (the interpolation idiom of ``mprl/util/util_matrix.py:195-227``)."""
    i0 = torch.clip(idx.floor().long(), 0, table.shape[0] - 2)
    w = idx - i0
    a, b = table[i0], table[i0 + 1]
    if table.ndim > 1:
        w = w[..., None]
    return torch.lerp(a, b, w)


class ProDMPOracle:
    def __init__(self, num_dof, num_basis, tau, alpha_phase, alpha, dt,
                 basis_bandwidth_factor, num_basis_outside=0, delay=0.0,
                 weights_scale=1.0, goal_scale=1.0, auto_scale_basis=True,
                 relative_goal=False, disable_goal=False,
                 disable_weights=False, pre_compute_length_factor=5,
                 dtype=torch.float32, cov_reg=1e-4):
        self.num_dof, self.num_basis = num_dof, num_basis
        self.tau, self.delay = float(tau), float(delay)
        self.alpha_phase, self.alpha, self.dt = float(alpha_phase), \
            float(alpha), float(dt)
        self.relative_goal = relative_goal
        self.disable_goal, self.disable_weights = disable_goal, disable_weights
        self.dtype, self.cov_reg = dtype, cov_reg
        self.num_basis_g = num_basis + 1
        self.factor = pre_compute_length_factor

        f64 = torch.float64
        # --- RBF centres / bandwidths in phase space
        nb, nbo = num_basis, num_basis_outside
        dist = self.tau / (nb - 2 * nbo - 1) if nb > 1 else self.tau
        c_t = torch.linspace(-nbo * dist + self.delay,
                             self.tau + nbo * dist + self.delay, nb, dtype=f64)
        c_p = torch.exp(-self.alpha_phase * (c_t - self.delay) / self.tau)
        if nb > 1:
            bw = torch.cat([c_p[1:] - c_p[:-1], c_p[-1:] - c_p[-2:-1]])
        else:
            bw = torch.ones(1, dtype=f64)
        bw = basis_bandwidth_factor / bw ** 2
        self.centers_p, self.bandwidth = c_p, bw

        # --- pre-compute grid in scaled time [0, factor]
        self.scaled_dt = self.dt / self.tau
        M = self.factor * int(round(1.0 / self.scaled_dt)) + 1
        s = torch.linspace(0, self.factor, M, dtype=f64)
        a = self.alpha
        y1 = torch.exp(-0.5 * a * s)
        y2 = s * y1
        dy1 = -0.5 * a * y1
        dy2 = -0.5 * a * y2 + y1
        q1 = (0.5 * a * s - 1) * torch.exp(0.5 * a * s) + 1
        q2 = 0.5 * a * (torch.exp(0.5 * a * s) - 1)
        x = torch.exp(-self.alpha_phase * s)              # s >= 0 on the grid
        rbf = torch.exp(-0.5 * (x[:, None] - c_p[None, :]) ** 2 * bw[None, :])
        if nb > 1:
            rbf = rbf / rbf.sum(-1, keepdim=True)
        e = torch.exp(0.5 * a * s) * x
        dp1 = (s * e)[:, None] * rbf
        dp2 = e[:, None] * rbf
        ds = s[1:] - s[:-1]
        p1 = torch.zeros_like(dp1)
        p2 = torch.zeros_like(dp2)
        p1[1:] = torch.cumsum(0.5 * (dp1[1:] + dp1[:-1]) * ds[:, None], 0)
        p2[1:] = torch.cumsum(0.5 * (dp2[1:] + dp2[:-1]) * ds[:, None], 0)
        pos_w = p2 * y2[:, None] - p1 * y1[:, None]
        vel_w = p2 * dy2[:, None] - p1 * dy1[:, None]
        pos_g = q2 * y2 - q1 * y1
        vel_g = q2 * dy2 - q1 * dy1
        pc_pos = torch.cat([pos_w, pos_g[:, None]], -1)
        pc_vel = torch.cat([vel_w, vel_g[:, None]], -1)

        # --- weight / goal scaling folded into the tables
        scale = torch.ones(self.num_basis_g, dtype=f64)
        if auto_scale_basis:
            scale = 1.0 / pc_pos.abs().max(dim=0).values
        scale[:-1] *= weights_scale
        scale[-1] *= goal_scale
        self.scale = scale.to(dtype)
        self.pc_pos = pc_pos.to(dtype)        # unscaled tables (as stored)
        self.pc_vel = pc_vel.to(dtype)
        self.y1, self.y2 = y1.to(dtype), y2.to(dtype)
        self.dy1, self.dy2 = dy1.to(dtype), dy2.to(dtype)
        self.num_pc = M

    # ---- parameter layout ------------------------------------------------
    @property
    def num_params(self):
        n = 0
        if not self.disable_weights:
            n += self.num_basis
        if not self.disable_goal:
            n += 1
        return n * self.num_dof

    def _pad(self, params):
        """[..., dof*n] -> [..., dof, nb+1] with zeros for disabled parts."""
        p = params.reshape(*params.shape[:-1], self.num_dof, -1)
        if self.disable_weights:
            p = torch.cat([p.new_zeros(*p.shape[:-1], self.num_basis), p], -1)
        if self.disable_goal:
            p = torch.cat([p, p.new_zeros(*p.shape[:-1], 1)], -1)
        return p

    # ---- table queries -----------------------------------------------------
    def _index(self, times):
        s = torch.clip((times - self.delay) / self.tau, min=0)
        assert s.max() <= self.factor, "time beyond the pre-computed range"
        return s / self.scaled_dt

    def _tables_at(self, times):
        idx = self._index(times)
        g = lambda tab: torch_lerp_rows(tab, idx)
        return (g(self.y1), g(self.y2), g(self.dy1), g(self.dy2),
                g(self.pc_pos) * self.scale, g(self.pc_vel) * self.scale,
                g(self.pc_pos)[..., -1], g(self.pc_vel)[..., -1])

    def basis_terms(self, times, init_time):
        """times [*, T], init_time [*] -> xi1..xi4 [*, T], H_pos, H_vel
        [*, T, nb+1] (scaled), Hg_pos, Hg_vel [*, T] (unscaled goal basis used
        by ``relative_goal``)."""
        y1, y2, dy1, dy2, P, V, Pg, Vg = self._tables_at(times)
        y1i, y2i, dy1i, dy2i, Pi, Vi, Pgi, Vgi = \
            self._tables_at(init_time[..., None])
        y1i, y2i, dy1i, dy2i = (v.squeeze(-1) for v in (y1i, y2i, dy1i, dy2i))
        Pi, Vi, Pgi, Vgi = Pi.squeeze(-2), Vi.squeeze(-2), \
            Pgi.squeeze(-1), Vgi.squeeze(-1)
        det = y1i * dy2i - y2i * dy1i
        e = lambda c, v: c[..., None] * v
        xi1 = e(dy2i / det, y1) - e(dy1i / det, y2)
        xi2 = e(y1i / det, y2) - e(y2i / det, y1)
        xi3 = e(dy2i / det, dy1) - e(dy1i / det, dy2)
        xi4 = e(y1i / det, dy2) - e(y2i / det, dy1)
        H_pos = P - xi1[..., None] * Pi[..., None, :] \
            - xi2[..., None] * Vi[..., None, :]
        H_vel = V - xi3[..., None] * Pi[..., None, :] \
            - xi4[..., None] * Vi[..., None, :]
        Hg_pos = Pg - xi1 * Pgi[..., None] - xi2 * Vgi[..., None]
        Hg_vel = Vg - xi3 * Pgi[..., None] - xi4 * Vgi[..., None]
        return xi1, xi2, xi3, xi4, H_pos, H_vel, Hg_pos, Hg_vel

    # ---- trajectories ------------------------------------------------------
    def traj(self, times, params, init_time, init_pos, init_vel):
        """pos, vel [*, T, dof] for parameters [*, dof*(nb+1)]."""
        xi1, xi2, xi3, xi4, Hp, Hv, Hgp, Hgv = \
            self.basis_terms(times, init_time)
        th = self._pad(params)                               # [*, dof, nbg]
        pos = torch.einsum('...tb,...db->...td', Hp, th)
        vel = torch.einsum('...tb,...db->...td', Hv, th)
        pos = pos + xi1[..., None] * init_pos[..., None, :] \
            + xi2[..., None] * (init_vel * self.tau)[..., None, :]
        vel = vel + xi3[..., None] * init_pos[..., None, :] \
            + xi4[..., None] * (init_vel * self.tau)[..., None, :]
        if self.relative_goal:
            pos = pos + Hgp[..., None] * init_pos[..., None, :]
            vel = vel + Hgv[..., None] * init_pos[..., None, :]
        return pos, vel / self.tau

    def sample_trajectories(self, times, params, params_L, init_time,
                            init_pos, init_vel, eps):
        """w = mean + L eps, then the trajectory (temporal_correlated_policy.py
        :76-85 with num_smp=1 squeezed; the noise is passed in explicitly)."""
        w = params + torch.einsum('...ij,...j->...i', params_L, eps)
        return self.traj(times, w, init_time, init_pos, init_vel)

    # ---- pair-wise distribution ---------------------------------------------
    def pos_H_multi(self, times, init_time):
        """Block H [*, dof*T, dof*nbg]: row d*T + j, col d*nbg + b."""
        Hp = self.basis_terms(times, init_time)[4]
        T = Hp.shape[-2]
        D, B = self.num_dof, self.num_basis_g
        H = Hp.new_zeros(*Hp.shape[:-2], D * T, D * B)
        for d in range(D):
            H[..., d * T:(d + 1) * T, d * B:(d + 1) * B] = Hp
        return H

    def traj_pos_flat(self, times, params, init_time, init_pos, init_vel):
        """dof-major flat positions [*, dof*T] (flat_shape=True)."""
        pos = self.traj(times, params, init_time, init_pos, init_vel)[0]
        return pos.transpose(-1, -2).reshape(*pos.shape[:-2], -1)

    def traj_pos_cov(self, times, params_L, init_time):
        assert not (self.disable_goal or self.disable_weights)
        H = self.pos_H_multi(times, init_time)
        cov = torch.einsum('...ij,...kj->...ik', params_L, params_L)
        out = torch.einsum('...ik,...kl,...jl->...ij', H, cov, H)
        return out + self.cov_reg * torch.eye(H.shape[-2], dtype=out.dtype)


def pair_log_prob(mp, smp_traj, params_mean, params_L, times, init_time,
                  init_pos, init_vel, pred_pairs):
    """temporal_correlated_policy.py:145-203: per (env, pair) Gaussian over the
    2*dof positions at the two pair times -> [N, P]."""
    P = pred_pairs.shape[0]
    D = mp.num_dof
    mean_e = params_mean[:, None, :].expand(-1, P, -1)
    L_e = params_L[:, None, :, :].expand(-1, P, -1, -1)
    time_pairs = times[:, pred_pairs]                        # [N, P, 2]
    t0_e = init_time[:, None].expand(-1, P)
    p0_e = init_pos[:, None, :].expand(-1, P, -1)
    v0_e = init_vel[:, None, :].expand(-1, P, -1)
    smp_pos = smp_traj[..., pred_pairs, :D]                  # [N, P, 2, dof]
    smp_pos = smp_pos.transpose(-1, -2).reshape(*smp_pos.shape[:-2], -1)
    mu = mp.traj_pos_flat(time_pairs, mean_e, t0_e, p0_e, v0_e)
    cov = mp.traj_pos_cov(time_pairs, L_e, t0_e)
    return torch.distributions.MultivariateNormal(
        loc=mu, covariance_matrix=cov, validate_args=False).log_prob(smp_pos)
