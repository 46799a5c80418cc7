/* Aerosol optics is OUT OF SCOPE of the hot path (no BASELINE config uses it). The type
 * exists only so that Radiation_solver_shortwave::solve_gpu keeps the reference's argument list
 * (/root/reference/include_test/Radiation_solver.h:175-218); passing switch_aerosol_optics = true throws. */
#ifndef AEROSOL_OPTICS_H
#define AEROSOL_OPTICS_H
class Aerosol_concs_gpu {};
#endif
