"""Shared builders for the tests: synthetic problems (BASELINE.json configs) with oracle-made data."""
import copy
import os
import numpy as np

from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.structs import HMCPrior

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def start_sigma(mesh):
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    return np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])


def make_problem(name, obs=None, err=None):
    """(mesh, data, inv, m_eval): obs/err default to the golden file's (oracle forward of the true
    model + seeded noise) when present, else to a smooth synthetic."""
    mesh, data, sig_true = S.make_config(name)
    if obs is None:
        path = os.path.join(GOLDEN, f"{name}.npz")
        if os.path.exists(path):
            g = np.load(path)
            obs, err = g["obs"], g["err"]
        else:
            n = len(data.rxID)
            obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
            err = np.full(n, 1e-3)
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    return mesh, data, inv, m


def oracle_eval(mesh, data, inv, m, dense_dbc=False, keep=None):
    from oracle import hmcmt_oracle as O
    if not mesh.setup:
        O.setupTensorMesh2D(mesh)
    inv.strModel = np.asarray(m, dtype=float).copy()
    return O.compDataGradient(mesh, data, inv, HMCPrior(), dense_dbc, keep)


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


def ragged_problem(ny, nz, nfreq, npad_y, npad_z, nair):
    """A configuration whose sizes are multiples of nothing: a single or a few frequencies, one receiver exactly on a
    node, a quarter of the data masked out, one fixed earth cell, a rough model.  Returns (mesh, data, inv, m)."""
    mesh = S.make_mesh(ny, nz, npad_y=npad_y, npad_z=npad_z, nair=nair)
    yN = np.concatenate([[0.0], np.cumsum(mesh.yLen)]) - mesh.origin[0]
    rx = np.array([yN[ny // 2], yN[ny // 2] + 130.0, yN[ny // 2 + 2] - 40.0, -350.0])     # first one on a node
    data = S.make_data_layout(S.log_freqs(nfreq) if nfreq > 1 else [3.7], np.sort(rx))
    rng = np.random.default_rng(ny)
    keep = rng.random(len(data.rxID)) > 0.25                                              # drop a quarter of the data
    data.dataID = keep.copy()
    data.rxID, data.freqID, data.dtID = data.rxID[keep], data.freqID[keep], data.dtID[keep]
    n = int(keep.sum())
    obs = (0.02 + 0.01 * rng.standard_normal(n)) * np.where(data.dtID == 1, 1.0, -1.0) * (1 + 1j)
    err = np.full(n, 2e-3)
    nt = mesh.gridSize[1]
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nt - nair), 0.01)])
    mesh.sigma[ny * nair + 5] = 0.3                                                        # a fixed (inactive) earth cell
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR, 0.3], 0.0, 0.0, obs, err)
    m = np.log(0.01) + 0.4 * rng.standard_normal(len(inv.strModel))
    return mesh, data, inv, m


def gerr_split(grad, ref, inv, mesh, deep_rows=5):
    """(max error over the shallow cells, over the deepest `deep_rows` rows) relative to max|ref|: the reference's
    bottom-boundary sensitivity row is rounding noise at mid/high frequencies (MT1DSensitivity.jl:145-155, SURVEY
    App. B.7), which reaches the deepest rows of the gradient at the 1e-6 level in ANY correct evaluation."""
    ny, nt = mesh.gridSize
    deep = (inv.activeIdx // ny) >= nt - deep_rows
    d = np.abs(np.asarray(grad) - np.asarray(ref))
    sc = np.abs(ref).max()
    return float(d[~deep].max() / sc), float(d[deep].max() / sc if deep.any() else 0.0)


def cfg3_subset_problem(g):
    """The headline mesh with the 4-frequency subset of tests/golden/cfg3s.npz, and the full 16-frequency problem on
    the same observations.  Returns (mesh, data4, inv4, m, data16, inv16)."""
    mesh, data16, _ = S.make_config("cfg3")
    fidx = g["fidx"]
    data = S.make_data_layout(data16.freqs[fidx], data16.rxLoc[:, 0])
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, g["obs"], g["err"])
    inv16 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, g["obs16"], g["err16"])
    return mesh, data, inv, S.rough_state(len(inv.strModel)), data16, inv16


class OracleContext:
    """Test double with the HipContext compute interface, backed by the oracle."""

    def __init__(self, mesh, data, inv):
        from oracle import hmcmt_oracle as O
        self.O, self.mesh, self.data = O, copy.deepcopy(mesh), data
        self.inv = copy.deepcopy(inv)
        O.setupTensorMesh2D(self.mesh)
        self._cache = None
        self.ngrad = self.nfwd = 0

    def grad(self, m):
        self.ngrad += 1
        self.inv.strModel = np.asarray(m).copy()
        return self.O.compDataGradient(self.mesh, self.data, self.inv, HMCPrior(), False)

    def forward(self, m):
        self.nfwd += 1
        s = self.inv.bgModel.copy(); s[self.inv.activeIdx] += np.exp(m); self.mesh.sigma = s
        p, _ = self.O.MT2DFwdSolver(self.mesh, self.data)
        return p, self.O.compDataMisfit(p, self.inv)


def rhophase_problem(name="tiny"):
    """tiny (or cfg1) config with DataType Rho_Pha and the masked data set of tests/golden/<name>_rhophase.npz.
    Returns (mesh, data, inv, m, golden)."""
    g = np.load(os.path.join(GOLDEN, f"{name}_rhophase.npz"))
    mesh, dz, _ = S.make_config(name)
    data = S.make_rhophase_layout(dz.freqs, dz.rxLoc[:, 0])
    keep = g["keep"]
    data.dataID = keep.copy()
    data.rxID, data.freqID, data.dtID = data.rxID[keep], data.freqID[keep], data.dtID[keep]
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, g["obs"], g["err"])
    return mesh, data, inv, S.rough_state(len(inv.strModel)), g


# ---- the reference example whose data pin the forward half (oracle/pin/recover_dprism.py) ----
DPRISM_TRUE = [(4000.0, 8000.0, 1000.0, 2000.0, 10.0), (8000.0, 12000.0, 1000.0, 2000.0, 1000.0)]   # y0, y1, z0, z1 (m), Ohm-m


def dprism_generating_problem():
    """(mesh, data, obs, err) of examples/dprism3d with the model its data were generated from -- a 10 Ohm-m and a
    1000 Ohm-m prism in 100 Ohm-m, recovered from the file's noise-free imaginary parts -- and the generator's
    frequencies (the logarithmic grid rounded to six digits; the file prints five)."""
    from hmcmt2d_amd import fileio
    ex = os.path.join(GOLDEN, "examples", "dprism3d")
    mesh = fileio.readEMModel2D(os.path.join(ex, "dprism2d_G96x49.mod"))
    data, obs, err = fileio.readMT2DData(os.path.join(ex, "dprism2dobs.dat"))
    data.freqs = np.array([float("%.6g" % f) for f in 10.0 ** np.linspace(2, -2, len(data.freqs))])
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    yN = np.concatenate([[0.0], np.cumsum(mesh.yLen)]) - mesh.origin[0]
    zN = np.concatenate([[0.0], np.cumsum(mesh.zLen)]) - mesh.origin[1]
    sig = np.full((nz, ny), 0.01)
    sig[:nair] = S.SIG_AIR
    for y0, y1, z0, z1, rho in DPRISM_TRUE:
        sig[np.searchsorted(zN, z0 - 1):np.searchsorted(zN, z1 - 1), np.searchsorted(yN, y0 - 1):np.searchsorted(yN, y1 - 1)] = 1.0 / rho
    mesh.sigma = sig.reshape(-1)
    return mesh, data, obs, err


def assert_reproduces_dprism_file(pred, obs, err, slack=0.0):
    """The 902 data of dprism2dobs.dat against a forward response of the generating model: every imaginary part
    (noise-free in the file) equal to the last of its seven printed digits, the error column = 5 % of |Z|, the
    real parts (5 % noise, clipped at two standard deviations by the generator) statistically consistent."""
    digit = lambda x: 10.0 ** (np.floor(np.log10(np.abs(x))) - 6)
    u = (pred - obs).imag / digit(obs.imag)
    assert np.abs(u).max() <= 0.5 + slack, np.abs(u).max()                 # i.e. '%.6e' % pred.imag == the file's text
    assert 0.27 < np.sqrt(np.mean(u * u)) < 0.31                           # rounding alone: 1/sqrt(12) = 0.289
    assert np.abs(0.05 * np.abs(pred) / err - 1).max() < 1e-6              # (pins |Z|, i.e. the real parts too)
    z = (obs - pred).real / err
    assert abs(z.mean()) < 0.1 and 0.9 < z.std() < 1.0 and np.abs(z).max() < 2.0 + 1e-4
