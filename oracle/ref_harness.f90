!==============================================================================
! ref_harness.f90 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
!
! A small driver (our own code) that `use`s the REFERENCE's Fortran modules,
! compiled from where they lie under /root/reference by oracle/build_ref.sh,
! and dumps raw fp64 results so that the C restatement (oracle/ljmd_oracle.c)
! and the HIP path can be pinned against the real reference arithmetic.
! The resulting binary lives in oracle/_ref/ (git-ignored, never committed).
!
! Reference entry points exercised:
!   init_params / init_state            scripts/base/md_types.f90:105,175
!   compute_lj_potential_energy         scripts/physics/lj_potential_energy.f90:46
!   verlet_step                         scripts/physics/verlet.f90:41
!   minimum_image / wrap_positions      scripts/physics/geometry_pbc.f90:80,39
!   random_uniform                      scripts/base/random_numbers.f90:48
! The per-step unwrapped-coordinate update is the caller-side loop of
! scripts/md_simulation_program.f90:339-353, restated here because it lives in
! the reference's main program, not in a module.
!
! Usage:
!   ref_harness force <in.bin> <out.bin>
!   ref_harness traj  <in.bin> <nsteps> <out.bin>
!   ref_harness bench <n> <ncalls>
!   ref_harness kat
!   ref_harness ran3  <seed> <count> <out.bin>     count draws of the reference's random_uniform, raw fp64
!
! <in.bin>  (stream, little endian): int32 n; real64 L, dt, rc;
!           rx(n) ry(n) rz(n) vx(n) vy(n) vz(n)
! force out: epot d_epot dd_epot ; ax(n) ay(n) az(n)
! traj  out: (epot ekin d_epot dd_epot) at t=0, then after every step
!            [4*(nsteps+1) doubles]; then final rx ry rz rux ruy ruz vx vy vz
!            ax ay az [12*n doubles]
!==============================================================================
program ref_harness
  use define_precision,    only: dp_kind, int_kind
  use md_types,            only: sim_params, sim_state, init_params, init_state
  use lj_potential_energy, only: compute_lj_potential_energy
  use verlet,              only: verlet_step
  use geometry_pbc,        only: minimum_image, wrap_positions
  use random_numbers,      only: random_uniform
  implicit none

  character(len=512) :: mode, a1, a2, a3
  integer :: nargs

  nargs = command_argument_count()
  if (nargs < 1) call usage()
  call get_command_argument(1, mode)

  select case (trim(mode))
  case ('force')
    if (nargs /= 3) call usage()
    call get_command_argument(2, a1); call get_command_argument(3, a2)
    call run_force(trim(a1), trim(a2))
  case ('traj')
    if (nargs /= 4) call usage()
    call get_command_argument(2, a1); call get_command_argument(3, a2)
    call get_command_argument(4, a3)
    call run_traj(trim(a1), to_int(a2), trim(a3))
  case ('bench')
    if (nargs /= 3) call usage()
    call get_command_argument(2, a1); call get_command_argument(3, a2)
    call run_bench(to_int(a1), to_int(a2))
  case ('kat')
    call run_kat()
  case ('ran3')
    if (nargs /= 4) call usage()
    call get_command_argument(2, a1); call get_command_argument(3, a2)
    call get_command_argument(4, a3)
    call run_ran3(to_int(a1), to_int(a2), trim(a3))
  case default
    call usage()
  end select

contains

  subroutine usage()
    write(*,'(a)') 'usage: ref_harness force|traj|bench|kat|ran3 ...'
    stop 2
  end subroutine usage

  integer function to_int(s)
    character(len=*), intent(in) :: s
    read(s, *) to_int
  end function to_int

  subroutine load_case(fname, p, s)
    character(len=*), intent(in)    :: fname
    type(sim_params), intent(inout) :: p
    type(sim_state),  intent(inout) :: s
    integer(kind=int_kind) :: n
    real(kind=dp_kind)     :: box, dt, rc
    integer :: iu, ios
    iu = 71
    open(iu, file=fname, access='stream', form='unformatted', status='old', action='read', iostat=ios)
    if (ios /= 0) stop 'ref_harness: cannot open input'
    read(iu) n
    read(iu) box, dt, rc
    call init_params(p, n, box, dt, rc)
    call init_state(p, s)
    read(iu) s%rx, s%ry, s%rz, s%vx, s%vy, s%vz
    close(iu)
  end subroutine load_case

  subroutine run_force(fin, fout)
    character(len=*), intent(in) :: fin, fout
    type(sim_params) :: p
    type(sim_state)  :: s
    real(kind=dp_kind) :: epot, d_epot, dd_epot
    integer :: iu
    call load_case(fin, p, s)
    call compute_lj_potential_energy(p, s, epot, d_epot, dd_epot)
    iu = 72
    open(iu, file=fout, access='stream', form='unformatted', status='replace', action='write')
    write(iu) epot, d_epot, dd_epot
    write(iu) s%ax, s%ay, s%az
    close(iu)
  end subroutine run_force

  subroutine run_traj(fin, nsteps, fout)
    character(len=*), intent(in) :: fin, fout
    integer, intent(in) :: nsteps
    type(sim_params) :: p
    type(sim_state)  :: s
    real(kind=dp_kind), allocatable :: ux(:), uy(:), uz(:), px(:), py(:), pz(:)
    real(kind=dp_kind) :: epot, ekin, d_epot, dd_epot, dx, dy, dz
    integer :: iu, step
    integer(kind=int_kind) :: i

    call load_case(fin, p, s)
    allocate(ux(p%n), uy(p%n), uz(p%n), px(p%n), py(p%n), pz(p%n))
    ux = s%rx; uy = s%ry; uz = s%rz

    iu = 72
    open(iu, file=fout, access='stream', form='unformatted', status='replace', action='write')

    ! t = 0 : same calls and the same fused ekin sum as md_simulation_program.f90:236-240
    call compute_lj_potential_energy(p, s, epot, d_epot, dd_epot)
    ekin = 0.5d0 * sum(s%vx*s%vx + s%vy*s%vy + s%vz*s%vz)
    write(iu) epot, ekin, d_epot, dd_epot

    do step = 1, nsteps
      px = s%rx; py = s%ry; pz = s%rz
      call verlet_step(p, s, epot, ekin, d_epot, dd_epot)
      do i = 1, p%n            ! md_simulation_program.f90:339-353
        dx = s%rx(i) - px(i); dy = s%ry(i) - py(i); dz = s%rz(i) - pz(i)
        dx = dx - p%box_length * dnint(dx * p%inv_box_length)
        dy = dy - p%box_length * dnint(dy * p%inv_box_length)
        dz = dz - p%box_length * dnint(dz * p%inv_box_length)
        ux(i) = ux(i) + dx; uy(i) = uy(i) + dy; uz(i) = uz(i) + dz
      end do
      write(iu) epot, ekin, d_epot, dd_epot
    end do

    write(iu) s%rx, s%ry, s%rz, ux, uy, uz, s%vx, s%vy, s%vz, s%ax, s%ay, s%az
    close(iu)
  end subroutine run_traj

  ! Times ncalls force evaluations on a jittered simple-cubic configuration at
  ! rho = 0.8, rc = 0.49 L (the BASELINE workload shape).  Prints one line:
  !   n ncalls seconds_total pairs_per_second epot
  subroutine run_bench(n, ncalls)
    integer, intent(in) :: n, ncalls
    type(sim_params) :: p
    type(sim_state)  :: s
    real(kind=dp_kind) :: box, a, epot, d_epot, dd_epot, secs, pairs
    integer(kind=int_kind) :: m, i, ix, iy, iz, seed
    integer(kind=8) :: c0, c1, rate
    integer :: k

    box = (dble(n) / 0.8d0)**(1.d0/3.d0)
    m = int(ceiling(dble(n)**(1.d0/3.d0) - 1.d-9), kind=int_kind)
    a = box / dble(m)
    call init_params(p, int(n, kind=int_kind), box, 5.d-3, 0.49d0*box)
    call init_state(p, s)
    seed = -20240601_int_kind
    i = 0
    outer: do ix = 0, m-1
      do iy = 0, m-1
        do iz = 0, m-1
          i = i + 1
          if (i > p%n) exit outer
          s%rx(i) = (dble(ix) + 0.5d0 + 0.1d0*(random_uniform(seed) - 0.5d0)) * a
          s%ry(i) = (dble(iy) + 0.5d0 + 0.1d0*(random_uniform(seed) - 0.5d0)) * a
          s%rz(i) = (dble(iz) + 0.5d0 + 0.1d0*(random_uniform(seed) - 0.5d0)) * a
        end do
      end do
    end do outer

    call system_clock(c0, rate)
    do k = 1, ncalls
      call compute_lj_potential_energy(p, s, epot, d_epot, dd_epot)
    end do
    call system_clock(c1)
    secs  = dble(c1 - c0) / dble(rate)
    pairs = 0.5d0 * dble(n) * dble(n - 1) * dble(ncalls)
    write(*,'(i0,1x,i0,1x,es16.8,1x,es16.8,1x,es24.16)') n, ncalls, secs, pairs/secs, epot
  end subroutine run_bench

  subroutine run_ran3(seed0, count, fout)
    integer, intent(in) :: seed0, count
    character(len=*), intent(in) :: fout
    integer(kind=int_kind) :: seed
    real(kind=dp_kind) :: r
    integer :: iu, k
    seed = seed0
    iu = 73
    open(iu, file=fout, access='stream', form='unformatted', status='replace', action='write')
    do k = 1, count
      r = random_uniform(seed)
      write(iu) r
    end do
    close(iu)
  end subroutine run_ran3

  subroutine run_kat()
    real(kind=dp_kind) :: x(3), y(3), z(3)
    integer(kind=int_kind) :: seed
    integer :: k
    write(*,'(a,3(1x,es24.16))') 'mic', minimum_image(9.5d0, 10.d0, 0.1d0), &
         minimum_image(5.d0, 10.d0, 0.1d0), minimum_image(-5.d0, 10.d0, 0.1d0)
    write(*,'(a,3(1x,es24.16))') 'mic2', minimum_image(4.999999d0, 10.d0, 0.1d0), &
         minimum_image(-14.9d0, 10.d0, 0.1d0), minimum_image(25.d0, 10.d0, 0.1d0)
    x = (/ -0.4d0, 10.2d0, 20.7d0 /); y = (/ 9.8d0, 0.d0, 10.d0 /); z = (/ -10.d0, -1.d-20, 9.999999999999999d0 /)
    call wrap_positions(x, y, z, 10.d0)
    write(*,'(a,3(1x,es24.16))') 'wrapx', x
    write(*,'(a,3(1x,es24.16))') 'wrapy', y
    write(*,'(a,3(1x,es24.16))') 'wrapz', z
    seed = -12345_int_kind
    do k = 1, 8
      write(*,'(a,1x,es24.16)') 'ran3', random_uniform(seed)
    end do
  end subroutine run_kat

end program ref_harness
