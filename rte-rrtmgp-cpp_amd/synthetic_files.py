"""Synthetic k-distribution / cloud / atmosphere files in the layout of rrtmgp-data and of the reference's
rte_rrtmgp_input.nc (variable names and dimension order as read by /root/reference/src_test/Radiation_solver.cu:99-357 and
src_test/test_rte_rrtmgp.cu:239-325,557-571), written as RRXB containers (rrxio.py) for the C++ driver.

The file layout is derived from the kernel-layout KDist (synthetic.py) by inverting what Gas_optics_rrtmgp::init_abs_coeffs
does, and ADDS a gas ('co') that the host model does not provide together with minor contributors that depend on it, so the
constructor's reduction logic (drop absent gases, re-pack kminor, renumber key species) is exercised: the C++ class must
arrive at exactly the kernel-layout tables again."""
import os
import numpy as np

from . import rrxio

FILE_GASES = ["h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "n2"]       # 'co' is absent from the host model


def _to_file_gas(i_reduced):
    """reduced (kernel) gas index 1..7 -> index in FILE_GASES (1..8); 0 stays 0"""
    return 0 if i_reduced == 0 else (i_reduced if i_reduced < 5 else i_reduced + 1)


def write_kdist(path, kd):
    gpb = kd.ngpt // kd.nbnd
    rng = np.random.default_rng(77)
    nfg = len(FILE_GASES)

    key_species = np.zeros((kd.nbnd, 2, 2), dtype=np.int32)          # (bnd, atmos_layer, pair)
    for ib in range(kd.nbnd):
        g0 = kd.band_lims_gpt[ib, 0] - 1
        for ia in range(2):
            f = kd.gpoint_flavor[g0, ia] - 1
            key_species[ib, ia, 0] = _to_file_gas(kd.flavor[f, 0])
            key_species[ib, ia, 1] = _to_file_gas(kd.flavor[f, 1])

    vmr_ref = np.zeros((kd.ntemp, nfg+1, 2))                          # (temperature, absorber_ext, atmos_layer)
    for i_red in range(kd.ngas+1):
        vmr_ref[:, _to_file_gas(i_red), :] = kd.vmr_ref[:, i_red, :]
    vmr_ref[:, 5, :] = 1.23e-7                                        # the absent gas

    def minor(kmin, limits, swd, sbc, idx, idx_scal):
        n = limits.shape[0]
        ins = n // 2                                                  # an interval of the absent gas goes here
        names = [kd.gas_names[i-1] for i in idx]
        scal = [kd.gas_names[i-1] if i > 0 else "" for i in idx_scal]
        k_file = np.transpose(kmin, (2, 1, 0))                        # (temperature, mixing_fraction, contributors)
        junk = rng.uniform(1., 2., (kd.ntemp, kd.neta, gpb))
        k_file = np.concatenate([k_file[..., :ins*gpb], junk, k_file[..., ins*gpb:]], axis=2)
        names = names[:ins] + ["co"] + names[ins:]
        scal = scal[:ins] + ["co"] + scal[ins:]
        lims = np.concatenate([limits[:ins], limits[ins-1:ins] if ins > 0 else limits[:1], limits[ins:]], axis=0)
        swd2 = np.concatenate([swd[:ins], [1], swd[ins:]]).astype(np.int32)
        sbc2 = np.concatenate([sbc[:ins], [0], sbc[ins:]]).astype(np.int32)
        start = (1 + gpb*np.arange(n+1)).astype(np.int32)
        return k_file, names, scal, lims.astype(np.int32), swd2, sbc2, start

    kl = minor(kd.kminor_lower, kd.minor_limits_gpt_lower, kd.minor_scales_with_density_lower, kd.scale_by_complement_lower,
               kd.idx_minor_lower, kd.idx_minor_scaling_lower)
    ku = minor(kd.kminor_upper, kd.minor_limits_gpt_upper, kd.minor_scales_with_density_upper, kd.scale_by_complement_upper,
               kd.idx_minor_upper, kd.idx_minor_scaling_upper)

    dims = dict(temperature=kd.ntemp, pressure=kd.npres, pressure_interp=kd.npres+1, absorber=nfg, absorber_ext=nfg+1,
                string_len=32, minor_absorber=nfg, mixing_fraction=kd.neta, atmos_layer=2, bnd=kd.nbnd, gpt=kd.ngpt, pair=2,
                minor_absorber_intervals_lower=kl[3].shape[0], minor_absorber_intervals_upper=ku[3].shape[0],
                contributors_lower=kl[0].shape[2], contributors_upper=ku[0].shape[2])
    S = rrxio.strings
    band_wn = np.stack([np.linspace(10, 3000, kd.nbnd), np.linspace(200, 3250, kd.nbnd)], axis=1)
    v = {
        "gas_names": (S(FILE_GASES), ["absorber", "string_len"]),
        "key_species": (key_species, ["bnd", "atmos_layer", "pair"]),
        "bnd_limits_wavenumber": (band_wn, ["bnd", "pair"]),
        "bnd_limits_gpt": (kd.band_lims_gpt.astype(np.int32), ["bnd", "pair"]),
        "press_ref": (np.exp(kd.press_ref_log), ["pressure"]),
        "temp_ref": (kd.temp_ref, ["temperature"]),
        "absorption_coefficient_ref_P": (np.array(1013.), []),
        "absorption_coefficient_ref_T": (np.array(296.), []),
        "press_ref_trop": (np.array(np.exp(kd.press_ref_trop_log)), []),
        "kminor_lower": (kl[0], ["temperature", "mixing_fraction", "contributors_lower"]),
        "kminor_upper": (ku[0], ["temperature", "mixing_fraction", "contributors_upper"]),
        "gas_minor": (S(FILE_GASES), ["minor_absorber", "string_len"]),
        "identifier_minor": (S(FILE_GASES), ["minor_absorber", "string_len"]),
        "minor_gases_lower": (S(kl[1]), ["minor_absorber_intervals_lower", "string_len"]),
        "minor_gases_upper": (S(ku[1]), ["minor_absorber_intervals_upper", "string_len"]),
        "scaling_gas_lower": (S(kl[2]), ["minor_absorber_intervals_lower", "string_len"]),
        "scaling_gas_upper": (S(ku[2]), ["minor_absorber_intervals_upper", "string_len"]),
        "minor_limits_gpt_lower": (kl[3], ["minor_absorber_intervals_lower", "pair"]),
        "minor_limits_gpt_upper": (ku[3], ["minor_absorber_intervals_upper", "pair"]),
        "minor_scales_with_density_lower": (kl[4], ["minor_absorber_intervals_lower"]),
        "minor_scales_with_density_upper": (ku[4], ["minor_absorber_intervals_upper"]),
        "scale_by_complement_lower": (kl[5], ["minor_absorber_intervals_lower"]),
        "scale_by_complement_upper": (ku[5], ["minor_absorber_intervals_upper"]),
        "kminor_start_lower": (kl[6], ["minor_absorber_intervals_lower"]),
        "kminor_start_upper": (ku[6], ["minor_absorber_intervals_upper"]),
        "vmr_ref": (vmr_ref, ["temperature", "absorber_ext", "atmos_layer"]),
        "kmajor": (np.transpose(kd.kmajor, (3, 1, 2, 0)), ["temperature", "pressure_interp", "mixing_fraction", "gpt"]),
    }
    if kd.kind == "lw":
        dims["temperature_Planck"] = kd.nPlanckTemp
        v["totplnk"] = (kd.totplnk, ["bnd", "temperature_Planck"])
        v["plank_fraction"] = (np.transpose(kd.planck_frac, (3, 1, 2, 0)), ["temperature", "pressure_interp", "mixing_fraction", "gpt"])
    else:
        v["rayl_lower"] = (np.transpose(kd.krayl[0], (2, 1, 0)), ["temperature", "mixing_fraction", "gpt"])
        v["rayl_upper"] = (np.transpose(kd.krayl[1], (2, 1, 0)), ["temperature", "mixing_fraction", "gpt"])
        v["solar_source_quiet"] = (kd.solar_source, ["gpt"])
        v["solar_source_facular"] = (np.zeros(kd.ngpt), ["gpt"])
        v["solar_source_sunspot"] = (np.zeros(kd.ngpt), ["gpt"])
        v["tsi_default"] = (np.array(kd.solar_source.sum()), [])
        v["mg_default"] = (np.array(0.1495954), [])
        v["sb_default"] = (np.array(0.00066696), [])
    rrxio.write(path, dims, v)


def write_cloud_lut(path, lut):
    nbnd = lut["lut_extliq"].shape[0]
    dims = dict(nband=nbnd, nrghice=3, nsize_liq=lut["nsize_liq"], nsize_ice=lut["nsize_ice"], pair=2)
    def ice(a):
        out = np.stack([a*0.5, a, a*2.0], axis=0)                     # roughness category 2 (index 1) is the one in use
        return out
    v = {"bnd_limits_wavenumber": (np.stack([np.linspace(10, 3000, nbnd), np.linspace(200, 3250, nbnd)], axis=1), ["nband", "pair"])}
    for k in ("radliq_lwr", "radliq_upr", "diamice_lwr", "diamice_upr"):
        v[k] = (np.array(float(lut[k])), [])
    v["radliq_fac"] = (np.array(0.0), []); v["diamice_fac"] = (np.array(0.0), [])
    for k in ("lut_extliq", "lut_ssaliq", "lut_asyliq"):
        v[k] = (lut[k], ["nband", "nsize_liq"])
    for k in ("lut_extice", "lut_ssaice", "lut_asyice"):
        v[k] = (ice(lut[k]), ["nrghice", "nband", "nsize_ice"])
    rrxio.write(path, dims, v)


def write_input(path, atm, nbnd_lw, nbnd_sw):
    ncol, nlay = atm.ncol, atm.nlay
    dims = dict(x=ncol, y=1, lay=nlay, lev=nlay+1, band_lw=nbnd_lw, band_sw=nbnd_sw)
    f3 = lambda a: a.reshape(a.shape[0], 1, ncol)
    v = {"p_lay": (f3(atm.p_lay), ["lay", "y", "x"]), "t_lay": (f3(atm.t_lay), ["lay", "y", "x"]),
         "p_lev": (f3(atm.p_lev), ["lev", "y", "x"]), "t_lev": (f3(atm.t_lev), ["lev", "y", "x"]),
         "vmr_h2o": (f3(atm.vmr["h2o"]), ["lay", "y", "x"]), "vmr_o3": (atm.vmr["o3"][:, 0].copy(), ["lay"]),
         "emis_sfc": (atm.emis_sfc.reshape(1, ncol, nbnd_lw), ["y", "x", "band_lw"]),
         "t_sfc": (atm.t_sfc.reshape(1, ncol), ["y", "x"]),
         "mu0": (atm.mu0.reshape(1, ncol), ["y", "x"]),
         "sfc_alb_dir": (atm.sfc_alb_dir.reshape(1, ncol, nbnd_sw), ["y", "x", "band_sw"]),
         "sfc_alb_dif": (atm.sfc_alb_dif.reshape(1, ncol, nbnd_sw), ["y", "x", "band_sw"]),
         "tsi_scaling": (np.array(float(atm.tsi_scaling[0])), [])}
    for g in ("co2", "ch4", "n2o", "n2", "o2"):
        v["vmr_" + g] = (np.array(float(atm.vmr[g][0, 0])), [])
    if atm.lwp is not None:
        for k in ("lwp", "iwp", "rel", "dei"):
            v[k] = (f3(getattr(atm, k)), ["lay", "y", "x"])
    if atm.rh is not None:
        v["rh"] = (f3(atm.rh), ["lay", "y", "x"])
        for k, a in atm.aermr.items():
            v[k] = (f3(a), ["lay", "y", "x"]) if a.ndim == 2 else (a.copy(), ["lay"])
    rrxio.write(path, dims, v)


def write_aerosol_lut(path, lut):
    """aerosol_optics.nc with the variables load_and_init_aerosol_optics reads (Radiation_solver.cu:366-401)."""
    nphobic, nbnd = lut["mext_phobic"].shape
    nphilic, nhum = lut["mext_philic"].shape[:2]
    dims = dict(band_sw=nbnd, relative_humidity=nhum, hydrophilic=nphilic, hydrophobic=nphobic)
    v = {"relative_humidity2": (lut["rh_upper"], ["relative_humidity"])}
    for file_name, key in (("mass_ext_sw", "mext"), ("ssa_sw", "ssa"), ("asymmetry_sw", "g")):
        v[file_name + "_hydrophobic"] = (lut[key + "_phobic"], ["hydrophobic", "band_sw"])
        v[file_name + "_hydrophilic"] = (lut[key + "_philic"], ["hydrophilic", "relative_humidity", "band_sw"])
    rrxio.write(path, dims, v)


def write_case(directory, atm, kd_lw, kd_sw, lut_lw=None, lut_sw=None, lut_aerosol=None):
    """Everything the C++ driver expects in its working directory (file names of the reference's make_links.sh)."""
    os.makedirs(directory, exist_ok=True)
    write_kdist(os.path.join(directory, "coefficients_lw.nc"), kd_lw)
    write_kdist(os.path.join(directory, "coefficients_sw.nc"), kd_sw)
    if lut_lw is not None:
        write_cloud_lut(os.path.join(directory, "cloud_coefficients_lw.nc"), lut_lw)
        write_cloud_lut(os.path.join(directory, "cloud_coefficients_sw.nc"), lut_sw)
    if lut_aerosol is not None:
        write_aerosol_lut(os.path.join(directory, "aerosol_optics.nc"), lut_aerosol)
    write_input(os.path.join(directory, "rte_rrtmgp_input.nc"), atm, kd_lw.nbnd, kd_sw.nbnd)
