!==============================================================================
! md_init_host -- the O(N) host arithmetic of the initial-configuration program
! (scripts/md_initial_config_program.f90), shared by the GPU driver md_initial_config_gpu and
! the GPU-free replay tool md_init_replay (tests/test_init_host.py pins it byte for byte to the
! reference's rv_init.dat):
!   build_fcc_lattice                     :132-187  cells ix > iy > iz, 4 basis atoms per cell
!   assign_random_velocities              :196-209  vx, vy, vz per particle from random_uniform - 0.5
!   remove_center_of_mass_velocity        :218-234  sum(v) / dble(n), subtracted per component
!   rescale_velocities_to_target_energy   :244-266  scale = sqrt((E_target - epot) / ekin_old)
!   write_rv_init                         :275-290  two unformatted records
! Every expression keeps the reference's operand order; the `sum` intrinsic's order is the
! compiler's on both sides (same compiler, same flags).
!==============================================================================
module md_init_host
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state
  use random_numbers,   only: random_uniform
  implicit none
  private
  public :: build_fcc_lattice, assign_random_velocities, remove_center_of_mass_velocity
  public :: rescale_velocities_to_target_energy, write_rv_init

contains

  subroutine build_fcc_lattice(p, s)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    integer(kind=int_kind) :: ix, iy, iz, idx
    real(kind=dp_kind) :: a, x0, y0, z0
    a = p%box_length / dble(p%num_cells)
    idx = 0
    do ix = 0, p%num_cells - 1
      do iy = 0, p%num_cells - 1
        do iz = 0, p%num_cells - 1
          x0 = dble(ix) * a; y0 = dble(iy) * a; z0 = dble(iz) * a
          call place(x0,             y0,             z0)
          call place(x0,             y0 + 0.5d0 * a, z0 + 0.5d0 * a)
          call place(x0 + 0.5d0 * a, y0,             z0 + 0.5d0 * a)
          call place(x0 + 0.5d0 * a, y0 + 0.5d0 * a, z0)
        end do
      end do
    end do
    if (idx /= p%n) stop 'build_fcc_lattice(): unexpected particle count.'
  contains
    subroutine place(x, y, z)
      real(kind=dp_kind), intent(in) :: x, y, z
      idx = idx + 1
      s%rx(idx) = x; s%ry(idx) = y; s%rz(idx) = z
    end subroutine place
  end subroutine build_fcc_lattice

  subroutine assign_random_velocities(p, s, seed)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    integer(kind=int_kind), intent(inout) :: seed
    integer(kind=int_kind) :: i
    do i = 1, p%n
      s%vx(i) = random_uniform(seed) - 0.5d0
      s%vy(i) = random_uniform(seed) - 0.5d0
      s%vz(i) = random_uniform(seed) - 0.5d0
    end do
  end subroutine assign_random_velocities

  subroutine remove_center_of_mass_velocity(p, s)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    real(kind=dp_kind) :: vcm(3)
    vcm(1) = sum(s%vx(1:p%n)) / dble(p%n)
    vcm(2) = sum(s%vy(1:p%n)) / dble(p%n)
    vcm(3) = sum(s%vz(1:p%n)) / dble(p%n)
    s%vx(1:p%n) = s%vx(1:p%n) - vcm(1)
    s%vy(1:p%n) = s%vy(1:p%n) - vcm(2)
    s%vz(1:p%n) = s%vz(1:p%n) - vcm(3)
  end subroutine remove_center_of_mass_velocity

  subroutine rescale_velocities_to_target_energy(p, s, target_energy, epot)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    real(kind=dp_kind), intent(in) :: target_energy, epot
    real(kind=dp_kind) :: ekin_old, ekin_new, scale
    ekin_old = 0.5d0 * sum(s%vx * s%vx + s%vy * s%vy + s%vz * s%vz)
    ekin_new = target_energy - epot
    if (ekin_new <= 0.d0) stop 'rescale_velocities_to_target_energy(): target energy too low (zero or negative kinetic).'
    if (ekin_old <= 0.d0) stop 'rescale_velocities_to_target_energy(): ekin_old <= 0 (cannot rescale).'
    scale = sqrt(ekin_new / ekin_old)
    s%vx(1:p%n) = s%vx(1:p%n) * scale
    s%vy(1:p%n) = s%vy(1:p%n) * scale
    s%vz(1:p%n) = s%vz(1:p%n) * scale
  end subroutine rescale_velocities_to_target_energy

  subroutine write_rv_init(filename, p, s)
    character(len=*), intent(in) :: filename
    type(sim_params), intent(in) :: p
    type(sim_state), intent(in) :: s
    integer :: iu, ios
    open(newunit=iu, file=filename, form='unformatted', status='replace', action='write', iostat=ios)
    if (ios /= 0) stop 'write_rv_init(): cannot open output file.'
    write(iu) s%rx(1:p%n), s%ry(1:p%n), s%rz(1:p%n)
    write(iu) s%vx(1:p%n), s%vy(1:p%n), s%vz(1:p%n)
    close(iu)
  end subroutine write_rv_init

end module md_init_host
