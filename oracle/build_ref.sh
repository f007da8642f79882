#!/usr/bin/env bash
# build_ref.sh -- TEST INFRASTRUCTURE.  Compiles the REFERENCE Fortran sources
# where they lie under $LJMD_REFERENCE (default /root/reference) with amdflang
# and writes every output into oracle/_ref/ (git-ignored; travels to the GPU
# box as prebuilt binaries).  No reference source is copied into this repo.
#
# Compile order = the reference's build/one_run/compile_md_simulation.sh:41-62;
# flags = its -O2 (line 22) plus -ffp-contract=off so that the binary has the
# no-FMA arithmetic the reference's own compiler (gfortran -O2 on x86-64,
# which never contracts without -march=native) produces.
#
# Products:
#   _ref/ref_harness                 our harness (oracle/ref_harness.f90) + reference modules
#   _ref/md_initial_config_program   the reference's init program, unmodified
#   _ref/md_simulation_program       the reference's production program, unmodified
# If LJMD_SHIM_DIR (our drop-in Fortran shim modules) and LJMD_LIB_DIR
# (libljmd.so) are set, also:
#   _ref/md_simulation_program_gpu   reference main program + reference base/stats
#                                    modules, but lj_potential_energy/verlet
#                                    replaced by OUR shim -> the drop-in proof.
set -euo pipefail

HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LJMD_REFERENCE:-/root/reference}"
OUT="${HERE}/_ref"
FC="${FC:-amdflang}"

if [[ ! -d "${REF}/scripts/physics" ]]; then
  echo "[build_ref] reference not present at ${REF}; keeping prebuilt ${OUT} (if any)"
  exit 0
fi
if ! command -v "${FC}" >/dev/null 2>&1; then
  echo "[build_ref] ${FC} not found; reference is unbuildable here" >&2
  exit 1
fi

S="${REF}/scripts"
mkdir -p "${OUT}/obj" "${OUT}/mod"
FFLAGS=(-O2 -ffp-contract=off -module-dir "${OUT}/mod" -I "${OUT}/mod")

MODS=(base/define_precision base/random_numbers base/md_types base/read_input_files
      physics/geometry_pbc physics/lj_potential_energy physics/verlet physics/thermodynamic_coefs
      stats/stats_math stats/md_means stats/md_correlations)
OBJS=()
for m in "${MODS[@]}"; do
  o="${OUT}/obj/$(basename "${m}").o"
  "${FC}" "${FFLAGS[@]}" -c "${S}/${m}.f90" -o "${o}"
  OBJS+=("${o}")
done

"${FC}" "${FFLAGS[@]}" "${OBJS[@]}" "${HERE}/ref_harness.f90"            -o "${OUT}/ref_harness"
"${FC}" "${FFLAGS[@]}" "${OBJS[@]}" "${S}/md_initial_config_program.f90" -o "${OUT}/md_initial_config_program"
"${FC}" "${FFLAGS[@]}" "${OBJS[@]}" "${S}/md_simulation_program.f90"     -o "${OUT}/md_simulation_program"

if [[ -n "${LJMD_SHIM_DIR:-}" && -n "${LJMD_LIB_DIR:-}" ]]; then
  G="${OUT}/gpu"; mkdir -p "${G}/obj" "${G}/mod"
  GF=(-O2 -ffp-contract=off -module-dir "${G}/mod" -I "${G}/mod")
  GOBJS=()
  for m in base/define_precision base/random_numbers base/md_types base/read_input_files physics/geometry_pbc; do
    o="${G}/obj/$(basename "${m}").o"; "${FC}" "${GF[@]}" -c "${S}/${m}.f90" -o "${o}"; GOBJS+=("${o}")
  done
  for f in ljmd_c_api lj_potential_energy verlet; do
    o="${G}/obj/${f}.o"; "${FC}" "${GF[@]}" -c "${LJMD_SHIM_DIR}/${f}.f90" -o "${o}"; GOBJS+=("${o}")
  done
  for m in physics/thermodynamic_coefs stats/stats_math stats/md_means stats/md_correlations; do
    o="${G}/obj/$(basename "${m}").o"; "${FC}" "${GF[@]}" -c "${S}/${m}.f90" -o "${o}"; GOBJS+=("${o}")
  done
  "${FC}" "${GF[@]}" "${GOBJS[@]}" "${S}/md_simulation_program.f90" \
      -L"${LJMD_LIB_DIR}" -lljmd -Wl,-rpath,"${LJMD_LIB_DIR}" -o "${OUT}/md_simulation_program_gpu"
  "${FC}" "${GF[@]}" "${GOBJS[@]}" "${S}/md_initial_config_program.f90" \
      -L"${LJMD_LIB_DIR}" -lljmd -Wl,-rpath,"${LJMD_LIB_DIR}" -o "${OUT}/md_initial_config_program_gpu"
fi
echo "[build_ref] OK -> ${OUT}"
