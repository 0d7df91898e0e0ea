"""TEST INFRASTRUCTURE ONLY -- NumPy / SciPy restatement of the reference's density-map action extraction
(rl_env/WRSN.py:229-287 `density_map_to_action`, with the normalisation WRSN.step applies first, WRSN.py:293-296).
The charging spot comes from SciPy's L-BFGS-B exactly like the reference, so it is only a lower bar for the objective
value (the iterates depend on the SciPy version: parity of the spot is unpinned)."""
import numpy as np


def normalise(action, epsilon=1e-9):                         # WRSN.py:293-296
    action = np.array(action, dtype=np.float64)
    if not (np.all((action >= 0) & (action <= 1)) and np.isclose(np.sum(action), 1)):
        action = np.exp(action)
        action = action / (np.sum(action) + epsilon)
    return action


def objective(loc, node_xy, alive, energy, cs, threshold, rng, alpha, beta):     # WRSN.py:239-247 (positive sign)
    d = np.sqrt((node_xy[:, 0] - loc[0]) ** 2 + (node_xy[:, 1] - loc[1]) ** 2)
    w = np.where(alive, cs / np.where(alive, energy - threshold, 1.0), 0.0)
    return float(np.sum((d <= rng) * alive * w * alpha / (d + beta) ** 2))


def density_map_to_action(dmap, frame, node_xy, alive, energy, cs, threshold, rng, alpha, beta, with_search=True):
    G = dmap.shape[0]
    unit = 1.0 / G
    W, H = frame[1] - frame[0], frame[3] - frame[2]
    up = lambda p: np.array([p[0] * W + frame[0], p[1] * H + frame[2]])          # WRSN.py:91-93
    down = lambda p: np.array([(p[0] - frame[0]) / W, (p[1] - frame[2]) / H])    # WRSN.py:86-88
    mi = np.unravel_index(np.argmax(dmap), dmap.shape)                            # :234
    lower = up([(mi[0] + 0.5) * unit - rng / W, (mi[1] + 0.5) * unit - rng / H])  # :236
    upper = up([(mi[0] + 0.5) * unit + rng / W, (mi[1] + 0.5) * unit + rng / H])  # :237
    bounds = [(lower[0], upper[0]), (lower[1], upper[1])]
    start = [(lower[0] + upper[0]) / 2, (lower[1] + upper[1]) / 2]
    spot, val = np.array(start), None
    if with_search:
        from scipy.optimize import minimize
        f = lambda loc: -objective(loc, node_xy, alive, energy, cs, threshold, rng, alpha, beta)
        res = minimize(f, start, bounds=bounds, method="L-BFGS-B")                # :249
        spot, val = np.array(res.x), -float(res.fun)
    flat = np.copy(dmap).flatten()                                               # :276-285
    th = np.percentile(flat, 99.9)
    flat[flat < th] = 0
    prob = flat.reshape(dmap.shape)
    prob = prob / np.sum(prob)
    return {"cell": (int(mi[0]), int(mi[1])), "bounds": bounds, "spot": spot, "objective": val,
            "down": down(spot), "third": float(prob[mi[0]][mi[1]])}
