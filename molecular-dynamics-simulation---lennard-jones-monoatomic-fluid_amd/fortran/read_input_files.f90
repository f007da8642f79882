!==============================================================================
! read_input_files -- parser of inputs/input_simulation_parameters.txt.
!
! Keeps the reference's file grammar verbatim (scripts/base/read_input_files.f90:87-171):
! blank lines and lines with '#' in column 1 are skipped; every other line is offered to
! the next missing numeric block through a list-directed read and silently skipped when
! that read fails (this is how the header-word lines disappear).  Block 1: k total_steps
! output_interval warmup_steps; block 2: dt L rc_over_L; block 3: target_total_energy.
! N = 4 k^3, rc = rc_over_L * L.  Same validation messages.
!==============================================================================
module read_input_files
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, init_params
  implicit none
  private
  public :: read_simulation_parameters

contains

  subroutine read_simulation_parameters(filename, params, total_steps, output_interval, &
                                        warmup_steps, rc_over_L, target_total_energy)
    character(len=*), intent(in)  :: filename
    type(sim_params), intent(out) :: params
    integer(kind=int_kind), intent(out) :: total_steps, output_interval, warmup_steps
    real(kind=dp_kind), intent(out) :: rc_over_L, target_total_energy

    integer :: unit_in, ios, stage
    character(len=512) :: line
    integer(kind=int_kind) :: k
    real(kind=dp_kind) :: dt, box

    total_steps = 0; output_interval = 0; warmup_steps = 0
    rc_over_L = 0.d0; target_total_energy = 0.d0
    k = 0; dt = 0.d0; box = 0.d0

    open(newunit=unit_in, file=filename, status='old', action='read', iostat=ios)
    if (ios /= 0) stop 'read_simulation_parameters(): cannot open input file.'

    stage = 1                      ! which numeric block is still missing
    do while (stage <= 3)
      read(unit_in, '(A)', iostat=ios) line
      if (ios /= 0) exit
      if (len_trim(line) == 0) cycle
      if (line(1:1) == '#') cycle
      select case (stage)
      case (1)
        read(line, *, iostat=ios) k, total_steps, output_interval, warmup_steps
        if (ios /= 0) cycle
        if (k <= 0)               stop 'read_simulation_parameters(): k must be > 0.'
        if (total_steps <= 0)     stop 'read_simulation_parameters(): total_steps must be > 0.'
        if (output_interval <= 0) stop 'read_simulation_parameters(): output_interval must be > 0.'
        if (warmup_steps < 0)     stop 'read_simulation_parameters(): warmup_steps must be >= 0.'
        stage = 2
      case (2)
        read(line, *, iostat=ios) dt, box, rc_over_L
        if (ios /= 0) cycle
        if (dt <= 0.d0)        stop 'read_simulation_parameters(): dt must be > 0.'
        if (box <= 0.d0)       stop 'read_simulation_parameters(): L must be > 0.'
        if (rc_over_L <= 0.d0) stop 'read_simulation_parameters(): rc_over_L must be > 0.'
        if (rc_over_L > 0.5d0) stop 'read_simulation_parameters(): rc_over_L must be <= 0.5 (minimum image).'
        stage = 3
      case (3)
        read(line, *, iostat=ios) target_total_energy
        if (ios /= 0) cycle
        stage = 4
      end select
    end do
    close(unit_in)

    if (stage <= 1) stop 'read_simulation_parameters(): missing Block 1 numeric line.'
    if (stage == 2) stop 'read_simulation_parameters(): missing Block 2 numeric line.'
    if (stage == 3) stop 'read_simulation_parameters(): missing Block 3 numeric line.'

    call init_params(params, 4_int_kind * k * k * k, box, dt, rc_over_L * box, num_cells=k)
  end subroutine read_simulation_parameters

end module read_input_files
