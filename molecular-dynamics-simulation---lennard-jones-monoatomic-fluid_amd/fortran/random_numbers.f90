!==============================================================================
! random_numbers -- the subtractive lagged generator the reference seeds its initial
! velocities with (scripts/base/random_numbers.f90:48-116: Knuth's algorithm in floating
! point, modulus 4e6, seed offset 1618033, lags 55/24).  Re-stated so that the GPU init
! driver draws the SAME velocity sequence, bit for bit: the integer state is scaled by the
! rounded constant 1/4e6 (`fac`, random_numbers.f90:61,112) -- a division by 4e6 is NOT the same
! double for 1.2e6 of the 4e6 states.  tests/test_init_host.py compares 10^4 draws of THIS module
! with the oracle (itself pinned to the reference's kat.json).
!==============================================================================
module random_numbers
  use define_precision, only: dp_kind, int_kind
  implicit none
  private
  public :: random_uniform

  real(kind=dp_kind), parameter :: modulus = 4.0d6, seed_offset = 1618033.d0
  real(kind=dp_kind), parameter :: to_unit = 1.d0 / modulus     ! `fac`: rounded once, then multiplied
  real(kind=dp_kind), save :: table(55)
  integer(kind=int_kind), save :: head = 0, tail = 0
  logical, save :: primed = .false.

contains

  function random_uniform(seed) result(r)
    integer(kind=int_kind), intent(inout) :: seed
    real(kind=dp_kind) :: r
    real(kind=dp_kind) :: cur, nxt
    integer(kind=int_kind) :: i, pos, pass

    if (seed <= 0 .or. .not. primed) then
      ! scatter the seed through the table in stride-21 order, then churn it four times
      primed = .true.
      cur = mod(abs(seed_offset - abs(seed)), modulus)
      table(55) = cur
      nxt = 1.d0
      do i = 1, 54
        pos = mod(21 * i, 55)
        table(pos) = nxt
        nxt = cur - nxt
        if (nxt < 0.d0) nxt = nxt + modulus
        cur = table(pos)
      end do
      do pass = 1, 4
        do i = 1, 55
          table(i) = table(i) - table(1 + mod(i + 30, 55))
          if (table(i) < 0.d0) table(i) = table(i) + modulus
        end do
      end do
      head = 0
      tail = 31
      seed = 1
    end if

    head = head + 1; if (head == 56) head = 1
    tail = tail + 1; if (tail == 56) tail = 1
    cur = table(head) - table(tail)
    if (cur < 0.d0) cur = cur + modulus
    table(head) = cur
    r = cur * to_unit
  end function random_uniform

end module random_numbers
