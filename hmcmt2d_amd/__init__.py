"""hmcmt2d_amd -- MI355X-native hot path of CUG-EMI/HMCMT2D behind the reference's own API.

Layout: `csrc/` (HIP kernels + C ABI, built into `libhmcmt_hip.so`), `lib.py` (ctypes binding),
`sampler.py` / `fileio.py` / `invsetup.py` / `structs.py` (host-side mirror of the reference's
Julia interface for this path), `synthetic.py` (BASELINE.json workloads).
Importing the package does not load the shared library; the first compute call does, and it raises
if the library or a HIP device is missing (there is no CPU fallback).
"""
from .structs import (TensorMesh2D, MTData, HMCPrior, HMCParameter, HMCStatus, InvDataModel,
                      initHMCPrior, initHMCParameter, initHMCStatus)
from .invsetup import setupInverseDataModel, setActiveElement, compDataWeightMat
from .fileio import (readEMModel2D, writeEMModel2D, readMT2DData, writeMT2DData, readstartupFile,
                     outputHMCSamples, getPosteriorModel)
from .sampler import (compDataGradient, compDataMisfit, getHamiltonian, proposeLeapfrog, proposeLeapfrogDevice,
                      runHMCSampler,
                      parallelHMCSampler, getKineticEnergy, getKineticGradient, getMomentumVector,
                      setMassMatrix, checkParameterBound, get_context, release_context)
from .lib import HipContext, HmcmtError, build_library

# aliases used by BASELINE.json's north_star / the user guide
runHMC = runHMCSampler
leapfrog = proposeLeapfrog
parallelHMCsampling = parallelHMCSampler
