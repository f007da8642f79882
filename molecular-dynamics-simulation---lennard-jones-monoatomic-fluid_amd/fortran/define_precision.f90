! define_precision -- numeric kinds of the hot path's type contract
! (same names and values as scripts/base/define_precision.f90:14,17: int32 / real64).
module define_precision
  use, intrinsic :: iso_c_binding, only: c_int32_t, c_double
  implicit none
  integer, parameter :: int_kind = c_int32_t   ! = selected_int_kind(9)
  integer, parameter :: dp_kind  = c_double    ! = selected_real_kind(15, 307)
end module define_precision
