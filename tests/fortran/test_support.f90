! test_support.f90 -- shared pieces of the Fortran test twins: HIP runtime bindings (the tests move data with
! plain hipMemcpy, no hipfort needed), launcher-environment queries and the closed-form expected values
! (same analytic oracle as tests/oracle_runner.py: value = gx + X*(gy + Y*gz) on zero-based global
! coordinates, -1 outside the interior; SURVEY.md section 8c items 6 and 7).
module test_support
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real64, error_unit
  use cudecomp
  implicit none

  integer, parameter :: WORLD_COMM = int(z'44000000')  ! MPI_COMM_WORLD of the MPICH ABI (cudecomp_mpi_compat.h)
  integer(c_int), parameter :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2

  interface
    function hipSetDevice(dev) bind(C, name="hipSetDevice") result(res)
      import
      integer(c_int), value :: dev
      integer(c_int) :: res
    end function hipSetDevice
    function hipGetDeviceCount(n) bind(C, name="hipGetDeviceCount") result(res)
      import
      integer(c_int) :: n
      integer(c_int) :: res
    end function hipGetDeviceCount
    function hipMemcpy(dst, src, bytes, kind) bind(C, name="hipMemcpy") result(res)
      import
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int), value :: kind
      integer(c_int) :: res
    end function hipMemcpy
    function hipMalloc(ptr, bytes) bind(C, name="hipMalloc") result(res)
      import
      type(c_ptr) :: ptr
      integer(c_size_t), value :: bytes
      integer(c_int) :: res
    end function hipMalloc
    function hipFree(ptr) bind(C, name="hipFree") result(res)
      import
      type(c_ptr), value :: ptr
      integer(c_int) :: res
    end function hipFree
    function hipDeviceSynchronize() bind(C, name="hipDeviceSynchronize") result(res)
      import
      integer(c_int) :: res
    end function hipDeviceSynchronize
    function hipStreamCreate(stream) bind(C, name="hipStreamCreate") result(res)
      import
      integer(c_intptr_t) :: stream
      integer(c_int) :: res
    end function hipStreamCreate
    function hipStreamSynchronize(stream) bind(C, name="hipStreamSynchronize") result(res)
      import
      integer(c_intptr_t), value :: stream
      integer(c_int) :: res
    end function hipStreamSynchronize
  end interface

  integer :: nfail = 0

contains

  integer function env_int(name, default)
    character(len=*), intent(in) :: name
    integer, intent(in) :: default
    character(len=64) :: buf
    integer :: stat, length
    env_int = default
    call get_environment_variable(name, buf, length, stat)
    if (stat == 0 .and. length > 0) read (buf(1:length), *) env_int
  end function env_int

  subroutine arg_int(i, v)
    integer, intent(in) :: i
    integer, intent(inout) :: v
    character(len=64) :: buf
    if (command_argument_count() >= i) then
      call get_command_argument(i, buf)
      read (buf, *) v
    end if
  end subroutine arg_int

  subroutine check(res, what)
    integer(c_int), intent(in) :: res
    character(len=*), intent(in) :: what
    if (res /= CUDECOMP_RESULT_SUCCESS) then
      write (error_unit, '(a,a,a,i0)') "FAILED: ", what, " returned ", res
      error stop 1
    end if
  end subroutine check

  subroutine hipcheck(res, what)
    integer(c_int), intent(in) :: res
    character(len=*), intent(in) :: what
    if (res /= 0) then
      write (error_unit, '(a,a,a,i0)') "HIP FAILED: ", what, " returned ", res
      error stop 1
    end if
  end subroutine hipcheck

  subroutine expect(cond, what)
    logical, intent(in) :: cond
    character(len=*), intent(in) :: what
    if (.not. cond) then
      nfail = nfail + 1
      write (error_unit, '(a,a)') "MISMATCH: ", what
    end if
  end subroutine expect

  ! Expected contents of a pencil described by `p` (one-based order/lo/hi as the Fortran API returns them):
  ! interior cells hold their global linear index, everything else `outside`.
  subroutine fill_expected(p, gdims, ref, outside)
    type(cudecompPencilInfo), intent(in) :: p
    integer, intent(in) :: gdims(3)
    real(real64), intent(out) :: ref(:)
    real(real64), intent(in) :: outside
    integer :: i0, i1, i2, g(3), l(3), k
    integer(int64) :: idx
    logical :: inside
    idx = 0
    do i2 = 1, p%shape(3)
      do i1 = 1, p%shape(2)
        do i0 = 1, p%shape(1)
          idx = idx + 1
          l = [i0, i1, i2]
          inside = .true.
          do k = 1, 3
            ! memory position k holds global axis order(k); halo_extents is in global axis order
            g(p%order(k)) = p%lo(k) + (l(k) - 1 - p%halo_extents(p%order(k)))
            if (g(p%order(k)) < p%lo(k) .or. g(p%order(k)) > p%hi(k)) inside = .false.
          end do
          if (inside) then
            ref(idx) = real((g(1) - 1) + gdims(1)*((g(2) - 1) + int(gdims(2), int64)*(g(3) - 1)), real64)
          else
            ref(idx) = outside
          end if
        end do
      end do
    end do
  end subroutine fill_expected

  ! Expected pencil after a complete halo update: halo cells hold the (periodically wrapped) neighbour's value,
  ! -1 where there is no neighbour; padding stays -1.
  subroutine fill_expected_halo(p, gdims, periods, ref)
    type(cudecompPencilInfo), intent(in) :: p
    integer, intent(in) :: gdims(3)
    logical, intent(in) :: periods(3)
    real(real64), intent(out) :: ref(:)
    integer :: i0, i1, i2, g(3), l(3), k, ax, h
    integer(int64) :: idx
    logical :: valid
    idx = 0
    do i2 = 1, p%shape(3)
      do i1 = 1, p%shape(2)
        do i0 = 1, p%shape(1)
          idx = idx + 1
          l = [i0, i1, i2]
          valid = .true.
          do k = 1, 3
            ax = p%order(k)
            h = p%halo_extents(ax)
            g(ax) = p%lo(k) + (l(k) - 1 - h)
            ! padding sits above the upper halo
            if (l(k) > (p%hi(k) - p%lo(k) + 1) + 2*h) valid = .false.
            if (g(ax) < 1 .or. g(ax) > gdims(ax)) then
              if (periods(ax)) then
                g(ax) = modulo(g(ax) - 1, gdims(ax)) + 1
              else
                valid = .false.
              end if
            end if
          end do
          if (valid) then
            ref(idx) = real((g(1) - 1) + gdims(1)*((g(2) - 1) + int(gdims(2), int64)*(g(3) - 1)), real64)
          else
            ref(idx) = -1.0_real64
          end if
        end do
      end do
    end do
  end subroutine fill_expected_halo

end module test_support
