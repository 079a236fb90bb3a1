"""Writes tests/golden/reference_crosscheck/dprism3d_m1.txt: the perturbed model m1 of tests/golden/example_dprism3d.npz
(m0 + 0.3 N(0,1), numpy default_rng(3): make_golden.py::make_example) as text, one value per line at 17 digits, for
julia/crosscheck_reference.jl -- Julia cannot re-seed numpy's generator.  Test infrastructure."""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(HERE, "example_dprism3d.npz"))
os.makedirs(os.path.join(HERE, "reference_crosscheck"), exist_ok=True)
with open(os.path.join(HERE, "reference_crosscheck", "dprism3d_m1.txt"), "w") as f:
    f.write("# m1 = m0 + 0.3 N(0,1) of tests/golden/example_dprism3d.npz (ln sigma of the 96 x 49 earth cells), one value per line\n")
    for v in g["m1"]:
        f.write(f"{v:.17e}\n")
print("wrote", len(g["m1"]), "values")
