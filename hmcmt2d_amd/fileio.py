"""Readers / writers of the reference's file formats (host-side compatibility contract).

Model file  `EM2DModelFile`   : readEMModel2D.jl:11-154, writeEMModel2D.jl:11-82
Data file   `MT2DData_1.0`    : readMT2DData.jl:14-179, writeMT2DData.jl:12-86
Startup file                  : HMCSampler/readstartupFile.jl:4-103
Sample / statistics outputs   : HMCSampler/HMCSampler.jl:605-642 (getPosteriorModel), :785-828
                                (outputHMCSamples)

Deliberate deviations from the reference parsers (they crash there): blank lines
are skipped instead of raising; `Model Type: log` applies exp element-wise.
"""
from __future__ import annotations

import time
import numpy as np

from .structs import MTData, TensorMesh2D, HMCPrior, initHMCPrior
from .invsetup import setupInverseDataModel


def _lines(path):
    with open(path, "r") as f:
        for raw in f:
            s = raw.strip()
            if not s or s[0] == "#":
                continue
            yield s


def _read_numbers(it, n):
    out = []
    while len(out) < n:
        out.extend(float(t) for t in next(it).split())
    return np.asarray(out[:n], dtype=np.float64)


# ----------------------------------------------------------------------------- model file
def readEMModel2D(modelfile: str) -> TensorMesh2D:
    it = _lines(modelfile)
    yLen = zLen = sigma = None
    airLayer = np.zeros(0)
    origin = np.zeros(2)
    resType = "Conductivity"
    ny = nz = nAir = 0
    for line in it:
        if "NY" in line:                       # keywords are case-sensitive, as in the reference
            ny = int(line.split()[-1]); yLen = _read_numbers(it, ny)
        elif "NZ" in line:
            nz = int(line.split()[-1]); zLen = _read_numbers(it, nz)
        elif "NAIR" in line:
            nAir = int(line.split()[-1]); airLayer = _read_numbers(it, nAir)
        elif "Resistivity Type" in line:
            resType = line.split()[-1]
        elif "Model Type" in line:
            modType = line.split()[-1]
            sigma = _read_numbers(it, ny * nz)
            if resType == "Resistivity":
                sigma = 1.0 / sigma
            if modType == "log":
                sigma = np.exp(sigma)
        elif "Origin" in line:
            t = line.split()
            origin = np.array([float(t[-2]), float(t[-1])])
    if yLen is None or zLen is None or sigma is None:
        raise ValueError(f"{modelfile}: incomplete model file")
    if nAir > 0:                                # air layers are listed bottom -> up (readEMModel2D.jl:135-145)
        zLen = np.concatenate([airLayer[::-1], zLen])
        origin = origin.copy(); origin[1] += airLayer.sum()
        sigma = np.concatenate([np.full(ny * nAir, 1e-8), sigma])
    return TensorMesh2D(yLen, zLen, airLayer, (ny, len(zLen)), origin, sigma)


def writeEMModel2D(modelfile: str, mesh: TensorMesh2D):
    ny, nz = len(mesh.yLen), len(mesh.zLen)
    nAir = len(mesh.airLayer)
    with open(modelfile, "w") as f:
        f.write("%-18s %s\n" % ("#Format:", "EMModel2DFile"))
        f.write("%-18s %s\n" % ("#Description:", "file generated in " + time.strftime("%c")))
        f.write("%-6s %4d\n" % ("NY:", ny))
        for i in range(ny):
            f.write("%10.2f" % mesh.yLen[i])
            if (i + 1) % 8 == 0:
                f.write("\n")
        if ny % 8 != 0:
            f.write("\n")
        if nAir > 0:
            f.write("%-6s %4d\n" % ("NAIR:", nAir))
            for i in range(nAir):
                f.write("%12.2f" % mesh.airLayer[i])
                if (i + 1) % 8 == 0:
                    f.write("\n")
            if nAir % 8 != 0:
                f.write("\n")
        f.write("%-6s %4d\n" % ("NZ:", nz - nAir))
        for i in range(nAir, nz):
            f.write("%10.2f" % mesh.zLen[i])
            if (i - nAir + 1) % 8 == 0:
                f.write("\n")
        if (nz - nAir) % 8 != 0:
            f.write("\n")
        sig = np.asarray(mesh.sigma)[ny * nAir:].reshape(nz - nAir, ny)
        f.write("%-18s %s\n" % ("Resistivity Type:", "Conductivity"))
        f.write("%-18s %s\n" % ("Model Type:", "Linear"))
        for k in range(nz - nAir):
            f.write("".join("%4.2e " % v for v in sig[k]) + "\n")
        origin = np.array(mesh.origin, dtype=float)
        if nAir > 0:
            origin[1] -= np.sum(mesh.airLayer)
        f.write("%-15s %4.2e %4.2e" % ("Origin (m):", origin[0], origin[1]))


# ----------------------------------------------------------------------------- data file
def readMT2DData(datafile: str):
    """Returns (MTData, obsData, dataErr)."""
    it = _lines(datafile)
    rxLoc = freqs = None
    dataType = None
    dataComp = []
    rxID = freqID = dtID = obs = err = None
    isComplex = False
    for line in it:
        if "Format" in line:
            continue
        if "Receiver Location" in line:
            nr = int(line.split()[-1])
            rxLoc = np.zeros((nr, 2))
            for i in range(nr):
                t = next(it).split(); rxLoc[i] = [float(t[0]), float(t[1])]
        elif "Frequencies" in line:
            nf = int(line.split()[-1])
            freqs = np.array([float(next(it).split()[0]) for _ in range(nf)])
        elif "DataType" in line:
            dataType = line.split()[-1]
            if dataType not in ("Impedance", "Rho_Pha"):
                raise ValueError(f"{dataType} is not supported.")
            isComplex = dataType == "Impedance"
        elif "DataComp" in line:
            nDt = int(line.split()[-1])
            dataComp = [next(it).strip() for _ in range(nDt)]
        elif "Data Block" in line:
            nData = int(line.split()[-1])
            freqID = np.zeros(nData, dtype=np.int64); rxID = np.zeros(nData, dtype=np.int64)
            dtID = np.zeros(nData, dtype=np.int64); err = np.zeros(nData)
            obs = np.zeros(nData, dtype=np.complex128 if isComplex else np.float64)
            for k in range(nData):
                t = next(it).split()
                freqID[k], rxID[k], dtID[k] = int(t[0]), int(t[1]), int(t[2])
                if isComplex:
                    obs[k] = float(t[3]) + 1j * float(t[4]); err[k] = float(t[5])
                else:
                    obs[k] = float(t[3]); err[k] = float(t[4])
    if rxLoc is None or freqs is None or obs is None:
        raise ValueError(f"{datafile}: incomplete data file")
    compTE = any("XY" in c for c in dataComp)
    compTM = any("YX" in c for c in dataComp)
    nDt, nr, nf = len(dataComp), rxLoc.shape[0], len(freqs)
    dataID = np.zeros((nf, nr, nDt), dtype=bool)          # linear index = dt + nDt*(rx + nr*freq)
    dataID[freqID - 1, rxID - 1, dtID - 1] = True
    info = MTData(rxLoc, freqs, dataType, dataComp, rxID, freqID, dtID, dataID.reshape(-1), compTE, compTM)
    return info, obs, err


def writeMT2DData(datafile: str, datInfo: MTData, predData, dataErr=None):
    predData = np.asarray(predData)
    if dataErr is None or len(dataErr) == 0:
        dataErr = np.abs(predData) * 0.03                  # writeMT2DData.jl:54
    elif len(dataErr) == 1:
        dataErr = np.abs(predData) * dataErr[0]
    with open(datafile, "w") as f:
        f.write("%-20s%s\n" % ("Format:", "MT2DData_1.0"))
        f.write("# %s\n" % ("file generated in " + time.strftime("%c")))
        nr = datInfo.rxLoc.shape[0]
        f.write("%-25s %4d\n" % ("Receiver Location (m):", nr))
        f.write("# %5s %5s\n" % ("Y", "Z"))
        for i in range(nr):
            f.write("%12.2f %12.2f\n" % (datInfo.rxLoc[i, 0], datInfo.rxLoc[i, 1]))
        f.write("%-20s%3d\n" % ("Frequencies (Hz):", len(datInfo.freqs)))
        for v in datInfo.freqs:
            f.write("%8.4e\n" % v)
        f.write("%-12s %12s\n" % ("DataType:", datInfo.dataType))
        f.write("%-15s %d\n" % ("DataComp:", len(datInfo.dataComp)))
        for c in datInfo.dataComp:
            f.write("%4s\n" % c)
        f.write("%-15s %d\n" % ("Data Block:", len(predData)))
        if np.iscomplexobj(predData):
            f.write("# %6s %6s %10s %10s %15s %12s\n" % ("FreqNo.", "RxNo.", "dataComp", "RealValue", "ImagValue", "Error"))
            for i in range(len(predData)):
                f.write("%5d %6d %8d %15.6e %15.6e %15.6e\n" % (datInfo.freqID[i], datInfo.rxID[i], datInfo.dtID[i],
                                                                predData[i].real, predData[i].imag, dataErr[i]))
        else:
            f.write("# %6s %6s %10s %10s %12s\n" % ("FreqNo.", "RxNo.", "dataComp", "RealValue", "Error"))
            for i in range(len(predData)):
                f.write("%5d %6d %8d %15.6e %15.6e\n" % (datInfo.freqID[i], datInfo.rxID[i], datInfo.dtID[i],
                                                         predData[i], dataErr[i]))


# ----------------------------------------------------------------------------- startup file
def readstartupFile(startupfile: str, basedir: str | None = None):
    """Returns (mtMesh, mtData, invParam, hmcprior) like the reference; file names are resolved
    relative to the startup file's directory unless `basedir` is given."""
    import os
    basedir = os.path.dirname(os.path.abspath(startupfile)) if basedir is None else basedir
    datafile = modelfile = None
    sigmin = sigmax = 0.0
    sigfix = [1e-8]
    prior: HMCPrior = initHMCPrior()
    for line in _lines(startupfile):
        t = line.split()
        if "datafile:" in line:
            datafile = t[-1]
        elif "modelfile:" in line:
            modelfile = t[-1]
        elif "burninsamples:" in line:
            prior.burninsamples = int(t[-1])
        elif "totalsamples:" in line:
            prior.totalsamples = int(t[-1])
        elif "fixedresistivity:" in line:
            # NOTE: unreachable in the reference because "resistivity:" matches first
            # (readstartupFile.jl:46,57); honoured here as the user guide documents it.
            sigfix.append(float(t[-1]))
        elif "resistivity:" in line:
            rhomin, rhomax = float(t[-3]), float(t[-2])
            sigmin, sigmax = 1.0 / rhomax, 1.0 / rhomin
            prior.sigBounds = [sigmin, sigmax]
            prior.sigmastd = (np.log(sigmax) - np.log(sigmin)) * 0.05
        elif "timeinterval:" in line:
            prior.dt = float(t[-1])
        elif "timestep:" in line:
            prior.timestep = [int(t[-2]), int(t[-1])]
        elif "linearsolver:" in line:
            prior.linearSolver = t[-1]
        elif "masstype:" in line:
            prior.massType = t[-1]
        elif "smoothparameter:" in line:
            prior.regParam = float(t[-1])
    if datafile is None or modelfile is None:
        raise ValueError("startup file must name a datafile and a modelfile")
    mtData, obsData, dataErr = readMT2DData(os.path.join(basedir, datafile))
    mtMesh = readEMModel2D(os.path.join(basedir, modelfile))
    invParam = setupInverseDataModel(mtMesh, sigfix, sigmin, sigmax, obsData, dataErr)
    return mtMesh, mtData, invParam, prior


# ----------------------------------------------------------------------------- outputs
def outputHMCSamples(hmcmodel, hmcstats, hmcdata, ichain=1, cputime=0.0, outdir="."):
    """hmcsamples_id$k.model / .data and hmcstatistics_id$k.log (HMCSampler.jl:785-828)."""
    import os
    nparam, nsamples = hmcmodel.shape
    with open(os.path.join(outdir, f"hmcsamples_id{ichain}.model"), "w") as f:
        for k in range(nsamples):
            f.write("".join("%8.4e " % v for v in hmcmodel[:, k]) + "\n")
    with open(os.path.join(outdir, f"hmcsamples_id{ichain}.data"), "w") as f:
        for k in range(nsamples + 1):
            f.write("".join("%12.4e %12.4e" % (v.real, v.imag) for v in hmcdata[:, k]) + "\n")
    hm = hmcstats.hmstats
    with open(os.path.join(outdir, f"hmcstatistics_id{ichain}.log"), "w") as f:
        f.write("Total elapsed time (s): %8.2f\n" % cputime)
        f.write("Totalsamples: %6d, nAccept: %6d, nReject: %6d\n" % (nsamples, hmcstats.nAccept, hmcstats.nReject))
        f.write("Starting status: dtMisfit=%8.1f,mNorm=%8.1f,KEnergy=%8.1f,HEnergy=%8.1f\n" % tuple(hm[:, 0]))
        f.write("iterNo   dtMisfit  mNorm   KEnergy  HEnergy  Accept \n")
        for k in range(1, nsamples + 1):
            f.write("%6d %8.4e %8.4e %8.4e %8.4e %2d\n" % (k, hm[0, k], hm[1, k], hm[2, k], hm[3, k],
                                                          int(hmcstats.acceptstats[k - 1])))


def getPosteriorModel(hmcmodel, mtMesh, invParam, hmcprior, outdir=".", write=True):
    """Posterior mean / std of ln(sigma) after burn-in; writes meanModel.model / stdModel.model
    (HMCSampler.jl:605-642).  Returns (meanModel, stdModel)."""
    import os
    burnin = hmcprior.burninsamples
    ens = hmcmodel[:, burnin:]
    meanModel = ens.mean(axis=1)
    var = (ens ** 2).mean(axis=1) - meanModel ** 2
    var[var < 0] = np.finfo(float).eps
    stdModel = np.sqrt(var)
    if write:
        sigma = invParam.bgModel.copy(); sigma[invParam.activeIdx] += np.exp(meanModel)
        mtMesh.sigma = sigma
        writeEMModel2D(os.path.join(outdir, "meanModel.model"), mtMesh)
        sigma = invParam.bgModel.copy(); sigma[invParam.activeIdx] += stdModel
        mtMesh.sigma = sigma
        writeEMModel2D(os.path.join(outdir, "stdModel.model"), mtMesh)
    return meanModel, stdModel
