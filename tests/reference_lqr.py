"""The reference's autopilot LQR design, repeated on OUR Jacobian (test infrastructure).

What the reference does (lib/FlightApps/design/c172/c172x_design.jl):
  * `linearize(vehicle, trim_params)` (lib/FlightPhysics/src/aircraftbase.jl:292-334): Cessna172Xv0(NED) trimmed in still air,
    then A, B, C, D = forward-difference Jacobians (FiniteDiff's default, lib/FlightPhysics/src/linearization.jl:100-126) of
    ẋ_ss = f(x_ss, u_ss), y_ss = h(x_ss, u_ss) with the 20 / 4 / 38 labelled variables of
    lib/FlightApps/src/c172/c172x/c172x.jl:332-448 (XStateSpace, UStateSpace, YStateSpace);
  * `get_design_model!` (c172x_design.jl:23-82): similarity transform x' = T x with T = C[x' labels, :], replacing
    (v_x, v_y, v_z, ω_eng) by (EAS, α, β, n_eng); longitudinal / lateral subsystems by label selection;
  * five LQR designs (te2te :149-216, tv2te :328-427, vh2te :433-536, ar2ar :563-618, φβ2ar :626-694):
    K = lqr(A, B, Q, R), integral augmentation where z is tracked with integral action, K_fwd = M₂₂ + K M₁₂ with M = [A B; C D]⁻¹.
  * the results at the 28 (EAS, h) nodes are the gain lookups the product ships (flight.jl_amd/data/c172x_ctl/*.h5).

Here `f_ode(x, u) -> (ẋ, y)` is any implementation of Cessna172Sv0(NED)'s f_ode! in the oracle's 27-row layout (rows 12-17 =
ψ θ φ ϕ λ h_e, 21-26 = ω_eb_b, v_eb_b); the Xv0 actuators — ṗ = (cmd − p)/τ, the airframe sees the positions
(c172x.jl:19-52,222-281) — are appended analytically. Differences are one-sided like the reference's (the aerodynamic and
engine maps are piecewise linear: a one-sided quotient stays on the side of a knot the reference's does), second order:
(−3 f(x) + 4 f(x + h) − f(x + 2h)) / 2h."""
import numpy as np

import reference_fixtures as rf

X_LABELS = ["p", "q", "r", "ψ", "θ", "φ", "v_x", "v_y", "v_z", "ϕ", "λ", "h", "α_filt", "β_filt", "ω_eng", "fuel",
            "thr_p", "ail_p", "ele_p", "rud_p"]                                        # XStateSpace, c172x.jl:332-345
U_LABELS = ["throttle_cmd", "aileron_cmd", "elevator_cmd", "rudder_cmd"]               # UStateSpace, :348-353
XP_LABELS = ["p", "q", "r", "ψ", "θ", "φ", "EAS", "α", "β", "ϕ", "λ", "h", "α_filt", "β_filt", "n_eng", "fuel",
             "thr_p", "ail_p", "ele_p", "rud_p"]                                       # c172x_design.jl:37-40
# where a state-space variable lives: ("x", row of the 27-row oracle layout in NED) or ("u", row of the 16 inputs)
_WHERE = {"p": ("x", 21), "q": ("x", 22), "r": ("x", 23), "ψ": ("x", 12), "θ": ("x", 13), "φ": ("x", 14),
          "v_x": ("x", 24), "v_y": ("x", 25), "v_z": ("x", 26), "ϕ": ("x", 15), "λ": ("x", 16), "h": ("x", 17),
          "α_filt": ("x", 0), "β_filt": ("x", 1), "ω_eng": ("x", 9), "fuel": ("x", 8),
          "thr_p": ("u", 0), "ail_p": ("u", 2), "ele_p": ("u", 3), "rud_p": ("u", 4)}   # FB_U_THROTTLE/AILERON/ELEVATOR/RUDDER
ACT_TAU = 1.0 / 20.0                                  # Actuator1, c172x.jl:21
OMEGA_RATED = 2700.0 * 2.0 * np.pi / 60.0             # PistonEngine ω_rated = 2700 rpm, c172s.jl:16-34 / piston.jl:220-250
REL_STEP = 1e-6                                       # × max(|x|, 1): truncation ~1e-12 f''', rounding ~1e-10 |f|


def _outputs(x, u, y):
    """the YStateSpace rows the designs read beyond the states themselves: EAS, α, β (c172x.jl:407-448)"""
    return np.stack([y[rf.Y_EAS], y[rf.Y_ALPHA], y[rf.Y_BETA]])


def _f_ss(xd):
    return np.stack([xd[_WHERE[k][1]] for k in X_LABELS[:16]])


def linearize(f_ode, x0, u0):
    """A [20, 20, n], B [20, 4, n], and the output rows (EAS, α, β) Cy [3, 20, n] of n trimmed aircraft x0 [27, n], u0 [16, n],
    in ONE batched call of f_ode (1 + 2 x 16 points per aircraft; the four actuator columns are differences in u)."""
    n = x0.shape[1]
    nd = 20
    X = np.repeat(x0[:, None, :], 1 + 2 * nd, axis=1)          # [27, 41, n]
    U = np.repeat(u0[:, None, :], 1 + 2 * nd, axis=1)
    H = np.zeros((nd, n))
    for j, lab in enumerate(X_LABELS):
        arr, row = (X, _WHERE[lab][1]) if _WHERE[lab][0] == "x" else (U, _WHERE[lab][1])
        base = arr[row, 0]
        h = REL_STEP * np.maximum(np.abs(base), 1.0)
        # exactly representable steps: (x + h) − x == h
        h = (base + h) - base
        H[j] = h
        arr[row, 1 + 2 * j] = base + h
        arr[row, 2 + 2 * j] = base + 2.0 * h
    xd, y = f_ode(X.reshape(27, -1), U.reshape(16, -1))
    f = np.vstack([_f_ss(xd), _outputs(None, None, y)]).reshape(19, 1 + 2 * nd, n)
    J = np.zeros((19, nd, n))
    for j in range(nd):
        J[:, j] = (-3.0 * f[:, 0] + 4.0 * f[:, 1 + 2 * j] - f[:, 2 + 2 * j]) / (2.0 * H[j])
    A = np.zeros((20, 20, n)); B = np.zeros((20, 4, n))
    A[:16] = J[:16]
    for k in range(4):                                          # ṗ = (cmd − p) / τ
        A[16 + k, 16 + k] = -1.0 / ACT_TAU
        B[16 + k, k] = 1.0 / ACT_TAU
    return A, B, J[16:]


def design_model(A, B, Cy):
    """get_design_model!(…; model = :full) for one aircraft: A', B' in the XP_LABELS coordinates (c172x_design.jl:36-58)"""
    T = np.eye(20)
    T[XP_LABELS.index("EAS")] = Cy[0]; T[XP_LABELS.index("α")] = Cy[1]; T[XP_LABELS.index("β")] = Cy[2]
    T[XP_LABELS.index("n_eng")] = 0.0; T[XP_LABELS.index("n_eng"), X_LABELS.index("ω_eng")] = 1.0 / OMEGA_RATED
    return T @ A @ np.linalg.inv(T), T @ B


def _sub(Ap, Bp, x_labels, u_labels):
    ix = [XP_LABELS.index(k) for k in x_labels]; iu = [U_LABELS.index(k) for k in u_labels]
    return Ap[np.ix_(ix, ix)], Bp[np.ix_(ix, iu)]


def _lqr(A, B, Q, R):
    from scipy.linalg import solve_continuous_are
    P = solve_continuous_are(A, B, Q, R)
    return np.linalg.solve(R, B.T @ P)


def _z_rows(x_labels, u_labels, z_labels):
    """C, D of the tracked outputs: each z is a state of the design model or a command (y = u there: C = 0, D = 1)"""
    C = np.zeros((len(z_labels), len(x_labels))); D = np.zeros((len(z_labels), len(u_labels)))
    for i, z in enumerate(z_labels):
        if z in x_labels:
            C[i, x_labels.index(z)] = 1.0
        else:
            D[i, u_labels.index(z)] = 1.0
    return C, D


def _k_fwd(A, B, C, D, K):
    n = A.shape[0]
    M = np.linalg.inv(np.block([[A, B], [C, D]]))
    return M[n:, n:] + K @ M[:n, n:]


LON_RED = ["q", "θ", "EAS", "α", "α_filt", "n_eng", "thr_p", "ele_p"]          # XLonRed
LON_FULL = ["q", "θ", "EAS", "α", "h", "α_filt", "n_eng", "thr_p", "ele_p"]    # XLonFull, c172x_design.jl:66
LAT_RED = ["p", "r", "φ", "EAS", "β", "β_filt", "ail_p", "rud_p"]              # XLatRed (ψ deleted, :559)
U_LON = ["throttle_cmd", "elevator_cmd"]
U_LAT = ["aileron_cmd", "rudder_cmd"]

DESIGNS = {
    # name: (x labels, u labels, z labels, Q diagonal on x, Q diagonal on the integrators (None: no integral action), R diagonal,
    #        K_fwd rule)
    "te2te": (LON_RED, U_LON, ["throttle_cmd", "elevator_cmd"], dict(q=1, θ=20, EAS=0.02), None, [100, 5], "inverse"),          # :173-188
    "tv2te": (LON_RED, U_LON, ["throttle_cmd", "EAS"], dict(q=20, EAS=0.3), [0.1, 0.01], [1, 0.1], "inverse"),                  # :362-382
    "vh2te": (LON_FULL, U_LON, ["EAS", "h"], dict(q=20, θ=100, EAS=0.06, h=0.04), [0.005, 0.001], [0.1, 0.05], "inverse"),      # :469-488
    "ar2ar": (LAT_RED, U_LAT, ["aileron_cmd", "rudder_cmd"], dict(r=0.1, φ=0.1), None, [0.1, 0.01], "identity"),                # :584-594
    "phibeta2ar": (LAT_RED, U_LAT, ["φ", "β"], dict(r=0.1, φ=2, β=5), None, [0.1, 0.03], "inverse"),                            # :646-675
}


def design(name, Ap, Bp):
    """K_fbk, K_fwd, K_int of one LQR lookup at one node, from the full design model"""
    xl, ul, zl, qx, qi, r, fwd = DESIGNS[name]
    A, B = _sub(Ap, Bp, xl, ul)
    C, D = _z_rows(xl, ul, zl)
    nx, nz = len(xl), len(zl)
    Q = np.diag([float(qx.get(k, 0.0)) for k in xl]); R = np.diag(np.array(r, dtype=np.float64))
    if qi is None:
        K = _lqr(A, B, Q, R)
        K_int = np.zeros((len(ul), nz))
    else:
        A_aug = np.block([[A, np.zeros((nx, nz))], [C, np.zeros((nz, nz))]])
        B_aug = np.vstack([B, D])
        Q_aug = np.diag(np.concatenate([np.diag(Q), np.array(qi, dtype=np.float64)]))
        K_aug = _lqr(A_aug, B_aug, Q_aug, R)
        K, K_int = K_aug[:, :nx], K_aug[:, nx:]
    K_fwd = np.eye(nz) if fwd == "identity" else _k_fwd(A, B, C, D, K)
    return K, K_fwd, K_int


def compare_all(A, B, Cy, log=print):
    """designs all five lookups at all 28 nodes from (A, B, Cy) [.., 28] and returns {name: {matrix: max deviation relative to
    the largest entry of that stored matrix over the nodes}}"""
    out = {}
    models = [design_model(A[..., k], B[..., k], Cy[..., k]) for k in range(28)]
    for name in DESIGNS:
        st = rf.stored(name)
        dev = {"K_fbk": 0.0, "K_fwd": 0.0, "K_int": 0.0}
        for k in range(28):
            got = dict(zip(("K_fbk", "K_fwd", "K_int"), design(name, *models[k])))
            for m in dev:
                want = st[m][..., k]
                scale = max(np.abs(want).max(), 1e-300) if np.abs(want).max() > 0 else 1.0
                dev[m] = max(dev[m], np.abs(got[m] - want).max() / scale)
        out[name] = dev
        log(f"{name:11s} gains redesigned from our Jacobian vs the reference's stored ones, max over 28 nodes (relative to each matrix's "
            f"largest entry): K_fbk {dev['K_fbk']:.2e}  K_fwd {dev['K_fwd']:.2e}  K_int {dev['K_int']:.2e}")
    return out
