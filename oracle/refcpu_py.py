"""TEST INFRASTRUCTURE ONLY (build container only). Python side of oracle/refcpu_runner.cpp: the reference's own CPU classes
Cloud_optics / Aerosol_optics (/root/reference/src/{Cloud_optics,Aerosol_optics,Optical_props,Gas_concs}.cpp, compiled
unmodified by `make -C oracle refcpu`) run on arrays handed over in a binary file. Used by oracle/make_golden.py only; the
tests read the committed fixtures, never this module's binaries."""
import os
import subprocess
import tempfile
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def exe(dtype):
    return os.path.join(_HERE, "_ref", "refcpu_dp" if np.dtype(dtype) == np.float64 else "refcpu_sp")


def have(dtype=np.float64):
    return os.path.exists(exe(dtype))


def _run(mode, dtype, ints, arrays, out_shapes):
    dtype = np.dtype(dtype)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            np.array([len(ints)] + list(ints), dtype="<i4").tofile(f)
            for a in arrays:
                np.ascontiguousarray(a, dtype=dtype).tofile(f)
        subprocess.run([exe(dtype), mode, fin, fout], check=True)
        flat = np.fromfile(fout, dtype=dtype)
    outs, pos = [], 0
    for s in out_shapes:
        n = int(np.prod(s))
        outs.append(flat[pos:pos+n].reshape(s).copy()); pos += n
    assert pos == flat.size, "refcpu wrote an unexpected number of values"
    return outs


def band_lims_wvn(nbnd):
    """Band limits only label the bands on these paths (Optical_props base class)."""
    e = np.linspace(10., 3250., nbnd + 1)
    return np.stack([e[:-1], e[1:]], axis=1)                     # numpy (nbnd, 2) = reference (2, nbnd)


def cloud_optics(dtype, lut, clwp, ciwp, reliq, deice, nrough=3):
    """lut: this repository's dictionary (ice tables already reduced to the roughness the reference selects, icergh = 2,
    src/Cloud_optics.cpp:59-68, stored as numpy (nbnd, nsize)). The runner is handed the THREE-roughness file layout with the
    table in category 2 and deliberately different values in 1 and 3, so the fixture also pins the selection.
    Returns tau, ssa, g of the two-stream variant and tau of the 1scl variant, each (nbnd, nlay, ncol)."""
    nlay, ncol = clwp.shape
    nbnd, nliq = lut["lut_extliq"].shape
    nice = lut["lut_extice"].shape[1]
    ice3 = lambda t: np.stack([t * (0.5 + 0.75*r) if r != 1 else t for r in range(nrough)], axis=0)   # (nrgh, nbnd, nsize)
    arrays = [np.array([lut["radliq_lwr"], lut["radliq_upr"], lut["diamice_lwr"], lut["diamice_upr"]]),
              band_lims_wvn(nbnd), lut["lut_extliq"], lut["lut_ssaliq"], lut["lut_asyliq"],
              ice3(lut["lut_extice"]), ice3(lut["lut_ssaice"]), ice3(lut["lut_asyice"]), clwp, ciwp, reliq, deice]
    shp = (nbnd, nlay, ncol)
    return _run("cloud", dtype, [ncol, nlay, nbnd, nliq, nice, nrough], arrays, [shp]*4)


def aerosol_optics(dtype, lut, aermr, rh, plev):
    """aermr: aermr01..11, each (nlay, ncol) or an (nlay,) profile (handed to the reference as a (1, nlay) array, which its own
    fill_aerosols_3d broadcasts, src/Aerosol_optics.cpp:152-164). Returns tau, ssa, g (nbnd, nlay, ncol)."""
    nlay, ncol = rh.shape
    nphobic, nbnd = lut["mext_phobic"].shape
    nphilic, nhum, _ = lut["mext_philic"].shape
    flags = [int(m.ndim == 1) for m in aermr]
    arrays = [band_lims_wvn(nbnd), lut["rh_upper"], lut["mext_phobic"], lut["ssa_phobic"], lut["g_phobic"],
              lut["mext_philic"], lut["ssa_philic"], lut["g_philic"], *aermr, rh, plev]
    return _run("aerosol", dtype, [ncol, nlay, nbnd, nhum, nphobic, nphilic] + flags, arrays, [(nbnd, nlay, ncol)]*3)
