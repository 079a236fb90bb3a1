"""Synthetic workloads of BASELINE.json `configs`.

Meshes follow the reference example HMCMT/examples/dprism3d/dprism2d_G96x49.mod:5-26:
uniform 200 m x 100 m core, 7 padding columns per side (400 ... 25600 m), 8 padding rows at
the bottom (200 ... 25600 m), 7 air layers 100/300/1e3/3e3/1e4/3e4/1e5 m (listed bottom -> up
in the file), origin at mid-mesh in y and at the earth surface in z.  Host-only numpy code.
"""
from __future__ import annotations

import numpy as np

from .structs import MTData, TensorMesh2D

AIR_LAYERS = np.array([100.0, 300.0, 1e3, 3e3, 1e4, 3e4, 1e5])   # bottom -> up
SIG_AIR = 1e-8                                                    # readEMModel2D.jl:140

# name -> (ny, nz_earth, nFreq, receivers (y0, y1, step))
CONFIGS = {
    "tiny":     dict(ny=12,  nz=8,   nfreq=3,  rx=(-500.0, 400.0, 300.0), npad_y=3, npad_z=3, nair=3),
    "cfg1":     dict(ny=96,  nz=49,  nfreq=4,  rx=(-8000.0, 8000.0, 400.0)),
    "cfg2":     dict(ny=50,  nz=25,  nfreq=8,  rx=(-3000.0, 3000.0, 300.0)),
    "cfg3":     dict(ny=200, nz=100, nfreq=16, rx=(-16000.0, 16000.0, 800.0)),
    "cfg5":     dict(ny=400, nz=200, nfreq=32, rx=(-32000.0, 32000.0, 800.0)),
}


def make_mesh(ny, nz, npad_y=7, npad_z=8, nair=7, dy=200.0, dz=100.0) -> TensorMesh2D:
    """Tensor mesh with `nz` earth rows (+ `nair` air rows prepended) and homogeneous 100 Ohm-m earth."""
    if ny <= 2 * npad_y or nz <= npad_z:
        raise ValueError("mesh too small for the requested padding")
    pad_y = dy * 2.0 ** np.arange(1, npad_y + 1)
    yLen = np.concatenate([pad_y[::-1], np.full(ny - 2 * npad_y, dy), pad_y])
    pad_z = dz * 2.0 ** np.arange(1, npad_z + 1)
    zEarth = np.concatenate([np.full(nz - npad_z, dz), pad_z])
    air = AIR_LAYERS[:nair].copy()
    zLen = np.concatenate([air[::-1], zEarth])
    nzt = len(zLen)
    origin = np.array([yLen.sum() / 2.0, air.sum()])
    sigma = np.concatenate([np.full(ny * nair, SIG_AIR), np.full(ny * nz, 0.01)])
    return TensorMesh2D(yLen, zLen, air, (ny, nzt), origin, sigma)


def make_data_layout(freqs, rx_y, rx_z=0.0) -> MTData:
    """Full impedance survey ZXY+ZYX, data sorted (freq, rx, comp) as the reference expects
    (SURVEY App. A 'Data ordering')."""
    freqs = np.asarray(freqs, dtype=np.float64)
    rx_y = np.asarray(rx_y, dtype=np.float64)
    nF, nR = len(freqs), len(rx_y)
    rxLoc = np.stack([rx_y, np.full(nR, rx_z)], axis=1)
    f, r, d = np.meshgrid(np.arange(1, nF + 1), np.arange(1, nR + 1), np.arange(1, 3), indexing="ij")
    return MTData(rxLoc, freqs, "Impedance", ["ZXY", "ZYX"], r.reshape(-1).astype(np.int64),
                  f.reshape(-1).astype(np.int64), d.reshape(-1).astype(np.int64),
                  np.ones(2 * nR * nF, dtype=bool), True, True)


def make_rhophase_layout(freqs, rx_y, rx_z=0.0) -> MTData:
    """The same survey as apparent resistivity + phase (DataType Rho_Pha, components RhoXY PhsXY RhoYX PhsYX in the
    order MT2DFwdSolver.jl:191-205 stacks them), data sorted (freq, rx, comp)."""
    freqs = np.asarray(freqs, dtype=np.float64)
    rx_y = np.asarray(rx_y, dtype=np.float64)
    nF, nR = len(freqs), len(rx_y)
    rxLoc = np.stack([rx_y, np.full(nR, rx_z)], axis=1)
    f, r, d = np.meshgrid(np.arange(1, nF + 1), np.arange(1, nR + 1), np.arange(1, 5), indexing="ij")
    return MTData(rxLoc, freqs, "Rho_Pha", ["RhoXY", "PhsXY", "RhoYX", "PhsYX"], r.reshape(-1).astype(np.int64),
                  f.reshape(-1).astype(np.int64), d.reshape(-1).astype(np.int64),
                  np.ones(4 * nR * nF, dtype=bool), True, True)


def log_freqs(n, fmax=100.0, fmin=0.01):
    return np.logspace(np.log10(fmax), np.log10(fmin), n)


def cell_centres(mesh: TensorMesh2D):
    yN = np.concatenate([[0.0], np.cumsum(mesh.yLen)]) - mesh.origin[0]
    zN = np.concatenate([[0.0], np.cumsum(mesh.zLen)]) - mesh.origin[1]
    return 0.5 * (yN[:-1] + yN[1:]), 0.5 * (zN[:-1] + zN[1:])


def true_model_sigma(mesh: TensorMesh2D, block=True):
    """100 Ohm-m over 10 Ohm-m at 2 km, optional 10 Ohm-m block y in [-1,1] km, z in [0.5,1.5] km."""
    ny, nzt = mesh.gridSize
    yc, zc = cell_centres(mesh)
    sig = np.full((nzt, ny), 0.01)
    sig[zc > 2000.0, :] = 0.1
    if block:
        sig[np.ix_((zc > 500.0) & (zc < 1500.0), np.abs(yc) < 1000.0)] = 0.1
    sig[zc < 0.0, :] = SIG_AIR
    return sig.reshape(-1)


def make_config(name):
    """Returns (mesh with homogeneous 100 Ohm-m start model, MTData layout, true sigma)."""
    c = CONFIGS[name]
    kw = {k: c[k] for k in ("npad_y", "npad_z", "nair") if k in c}
    mesh = make_mesh(c["ny"], c["nz"], **kw)
    y0, y1, st = c["rx"]
    rx_y = np.arange(y0, y1 + 0.5 * st, st)
    data = make_data_layout(log_freqs(c["nfreq"]), rx_y)
    return mesh, data, true_model_sigma(mesh, block=(name != "cfg1"))


def noisy_observations(pred, rel=0.03, seed=20250114):
    """obs = pred + rel*|pred|*(N(0,1)+iN(0,1))/sqrt(2), err = rel*|pred| (cf. writeMT2DData.jl:54)."""
    rng = np.random.default_rng(seed)
    amp = rel * np.abs(pred)
    noise = (rng.standard_normal(len(pred)) + 1j * rng.standard_normal(len(pred))) / np.sqrt(2.0)
    return pred + amp * noise, amp


def rough_state(nparam, seed=1, centre=np.log(0.01), std=0.3):
    """Evaluation state for timing: m = ln(0.01) + 0.3 N(0,1)."""
    return centre + std * np.random.default_rng(seed).standard_normal(nparam)
