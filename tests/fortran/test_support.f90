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

  ! one command line, split at blanks ("--name v1 v2 ..." options and single-letter flags, the reference's test CLI)
  integer, parameter :: MAXTOK = 96
  type :: cmdline
    integer :: n = 0
    character(len=48) :: tok(MAXTOK)
  end type cmdline

contains

  subroutine tokenize(line, c)
    character(len=*), intent(in) :: line
    type(cmdline), intent(out) :: c
    integer :: i, start
    logical :: in_tok
    c%n = 0
    in_tok = .false.
    start = 1
    do i = 1, len_trim(line) + 1
      if (i <= len_trim(line) .and. line(i:min(i, len(line))) /= ' ') then
        if (.not. in_tok) then
          in_tok = .true.
          start = i
        end if
      else if (in_tok) then
        in_tok = .false.
        if (c%n < MAXTOK) then
          c%n = c%n + 1
          c%tok(c%n) = line(start:i - 1)
        end if
      end if
    end do
  end subroutine tokenize

  integer function find_opt(c, name)
    type(cmdline), intent(in) :: c
    character(len=*), intent(in) :: name
    integer :: i
    find_opt = 0
    do i = 1, c%n
      if (trim(c%tok(i)) == name) find_opt = i
    end do
  end function find_opt

  ! the integers that follow option `name` (as many as vals holds); vals keeps its content when the option is absent
  subroutine opt_ints(c, name, vals)
    type(cmdline), intent(in) :: c
    character(len=*), intent(in) :: name
    integer, intent(inout) :: vals(:)
    integer :: i, k, stat, v
    i = find_opt(c, name)
    if (i == 0) return
    do k = 1, size(vals)
      if (i + k > c%n) exit
      read (c%tok(i + k), *, iostat=stat) v
      if (stat /= 0) exit
      vals(k) = v
    end do
  end subroutine opt_ints

  integer function opt_int(c, name, default)
    type(cmdline), intent(in) :: c
    character(len=*), intent(in) :: name
    integer, intent(in) :: default
    integer :: v(1)
    v(1) = default
    call opt_ints(c, name, v)
    opt_int = v(1)
  end function opt_int

  ! data type of this executable from its name (transpose_test_R32, ..._R64, ..._C32, ..._C64; default R64):
  ! 1 real32, 2 real64, 3 complex32, 4 complex64
  integer function dtype_from_program_name()
    character(len=512) :: name
    integer :: n
    call get_command_argument(0, name)
    n = len_trim(name)
    dtype_from_program_name = 2
    if (n >= 4) then
      select case (name(n - 3:n))
      case ("_R32"); dtype_from_program_name = 1
      case ("_R64"); dtype_from_program_name = 2
      case ("_C32"); dtype_from_program_name = 3
      case ("_C64"); dtype_from_program_name = 4
      end select
    end if
  end function dtype_from_program_name

  ! Verdict of a case over all ranks without MPI: every rank drops a file into a job directory under /dev/shm, rank 0
  ! collects them (the ranks of these tests share a node).  Returns the maximum over ranks on rank 0.
  integer function reduce_verdict(mine, case_index)
    integer, intent(in) :: mine, case_index
    integer :: rank, nranks, r, v, u, stat, worst, c0, c1, rate
    character(len=64) :: job
    character(len=256) :: dir, fname
    logical :: ex
    integer :: jl, js
    rank = env_int("RANK", env_int("PMI_RANK", env_int("OMPI_COMM_WORLD_RANK", 0)))
    nranks = env_int("WORLD_SIZE", env_int("PMI_SIZE", env_int("OMPI_COMM_WORLD_SIZE", 1)))
    reduce_verdict = mine
    if (nranks == 1) return
    call get_environment_variable("CUDECOMP_TEST_JOB", job, jl, js)
    if (js /= 0 .or. jl == 0) call get_environment_variable("CUDECOMP_BOOTSTRAP_PORT", job, jl, js)
    if (js /= 0 .or. jl == 0) call get_environment_variable("MASTER_PORT", job, jl, js)
    if (js /= 0 .or. jl == 0) then
      job = "job"
      jl = 3
    end if
    dir = "/dev/shm/cudecomp_fortran_"//job(1:jl)
    call execute_command_line("mkdir -p "//trim(dir))
    write (fname, '(a,a,i0,a,i0)') trim(dir), "/case", case_index, "_rank", rank
    open (newunit=u, file=trim(fname)//".tmp", status="replace", action="write")
    write (u, '(i0)') mine
    close (u)
    call rename_file(trim(fname)//".tmp", trim(fname))
    if (rank /= 0) return
    worst = mine
    call system_clock(c0, rate)
    do r = 0, nranks - 1
      write (fname, '(a,a,i0,a,i0)') trim(dir), "/case", case_index, "_rank", r
      do
        inquire (file=trim(fname), exist=ex)
        if (ex) then
          open (newunit=u, file=trim(fname), status="old", action="read", iostat=stat)
          if (stat == 0) then
            read (u, *, iostat=stat) v
            close (u, status="delete")
            if (stat == 0) then
              worst = max(worst, v)
              exit
            end if
          end if
        end if
        call system_clock(c1)
        if (real(c1 - c0)/real(rate) > 300.0) then
          worst = 1
          exit
        end if
      end do
    end do
    reduce_verdict = worst
  end function reduce_verdict

  subroutine rename_file(from, to)
    character(len=*), intent(in) :: from, to
    interface
      function c_rename(a, b) bind(C, name="rename") result(res)
        import
        character(kind=c_char), intent(in) :: a(*), b(*)
        integer(c_int) :: res
      end function c_rename
    end interface
    integer(c_int) :: res
    res = c_rename(trim(from)//c_null_char, trim(to)//c_null_char)
  end subroutine rename_file

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
