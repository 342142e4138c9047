"""Synthetic molecule batches with the shapes BASELINE.json / SURVEY.md section 8(d) prescribe (no datasets offline)."""
import numpy as np
import torch

QM9_Z = np.array([1, 6, 7, 8, 9])
QM9_P = np.array([0.51, 0.35, 0.06, 0.07, 0.01])


def _sample_points(rng, n, side, min_dist):
    pts = []
    while len(pts) < n:
        p = rng.uniform(0.0, side, size=3)
        if all(np.linalg.norm(p - q) >= min_dist for q in pts):
            pts.append(p)
    return np.stack(pts)


def qm9_like_batch(num_molecules, atoms_per_molecule=18, side=6.5, min_dist=0.9, seed=0):
    """B molecules x N atoms, positions uniform in a cube of edge `side` (Angstrom) with a minimum pair distance.
    side=6.5 gives ~200 directed edges / molecule at r=5 (the north-star point), side=5.0 gives ~277 (QM9 statistics).
    Returns dict(pos [B*N,3] f32, z [B*N] i64 atomic numbers, batch [B*N] i64, y [B] f32)."""
    rng = np.random.default_rng(seed)
    pos = np.concatenate([_sample_points(rng, atoms_per_molecule, side, min_dist) for _ in range(num_molecules)])
    z = rng.choice(QM9_Z, size=num_molecules * atoms_per_molecule, p=QM9_P)
    batch = np.repeat(np.arange(num_molecules), atoms_per_molecule)
    y = rng.standard_normal(num_molecules)
    return dict(pos=torch.from_numpy(pos.astype(np.float32)), z=torch.from_numpy(z.astype(np.int64)),
                batch=torch.from_numpy(batch.astype(np.int64)), y=torch.from_numpy(y.astype(np.float32)))


# a plausible aspirin geometry (Angstrom): 9 C, 4 O, 8 H  [SURVEY.md section 8(d): Z = 6x9, 8x4, 1x8]
_ASPIRIN_Z = [6] * 9 + [8] * 4 + [1] * 8
_ASPIRIN_POS = np.array([
    [1.24, 0.72, 0.0], [2.45, 0.02, 0.0], [2.45, -1.38, 0.0], [1.24, -2.08, 0.0], [0.03, -1.38, 0.0],
    [0.03, 0.02, 0.0], [-1.25, 0.77, 0.1], [-1.10, -3.35, 0.9], [-2.10, -4.30, 1.4],
    [-1.35, 1.98, 0.3], [-2.33, -0.02, -0.1], [-1.17, -2.10, 0.2], [-0.20, -3.65, 1.3],
    [1.23, 1.81, 0.0], [3.39, 0.56, 0.0], [3.39, -1.93, 0.0], [1.25, -3.17, 0.0], [-3.12, 0.53, 0.0],
    [-1.70, -5.25, 1.8], [-2.80, -4.55, 0.6], [-2.70, -3.85, 2.2]]) * 0.85  # compact: ~350 directed edges at r=5


def md17_aspirin_batch(num_frames, jitter=0.05, seed=0):
    rng = np.random.default_rng(seed)
    n = len(_ASPIRIN_Z)
    pos = np.concatenate([_ASPIRIN_POS + rng.normal(0.0, jitter, size=(n, 3)) for _ in range(num_frames)])
    z = np.tile(np.array(_ASPIRIN_Z), num_frames)
    batch = np.repeat(np.arange(num_frames), n)
    return dict(pos=torch.from_numpy(pos.astype(np.float32)), z=torch.from_numpy(z.astype(np.int64)),
                batch=torch.from_numpy(batch.astype(np.int64)), y=torch.from_numpy(rng.standard_normal(num_frames).astype(np.float32)),
                dy=torch.from_numpy(rng.standard_normal((num_frames * n, 3)).astype(np.float32)))
