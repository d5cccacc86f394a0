! transpose_test.f90 -- Fortran twin of tests/native/transpose_test.cpp: X->Y->Z->Y->X through the Fortran module on
! device buffers, every stage compared element-for-element with the closed-form pencil contents.  Same command line,
! test-file mode and output protocol as the reference's tests/fortran/transpose_test.f90, in FORTRAN conventions
! (--mem_order entries are 1..3), so the `*_fortran` configurations of the reference's tests/test_config.yaml drive it
! unchanged.  The data type comes from the executable's name (transpose_test_R32 / _R64 / _C32 / _C64 are links to
! this one program; R64 otherwise).
!
!   --gx --gy --gz N   --pr --pc N (0 0 = autotune)   --rank-order 0|1|2   --backend B (0 = autotune)
!   --acx --acy --acz 0|1   --gd a b c   --hex|--hey|--hez a b c   --pdx|--pdy|--pdz a b c   --mem_order 9 ints (1-based)
!   -o out of place   -m accepted, ignored   -f|--testfile FILE
program transpose_test
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real32, real64, error_unit
  use cudecomp
  use test_support
  implicit none

  type(cudecompHandle) :: handle
  integer :: rank, nranks, ndev, dtype_sel, dtype, wpe, i, ncases, res, nfailed, u, stat, argn
  integer(int64) :: es
  character(len=1024) :: line, arg, testfile, progname
  character(len=1024), allocatable :: cases(:)
  logical :: from_file
  integer :: c0, c1, rate

  rank = env_int("RANK", env_int("PMI_RANK", env_int("OMPI_COMM_WORLD_RANK", 0)))
  nranks = env_int("WORLD_SIZE", env_int("PMI_SIZE", env_int("OMPI_COMM_WORLD_SIZE", 1)))
  dtype_sel = dtype_from_program_name()
  select case (dtype_sel)
  case (1); dtype = CUDECOMP_FLOAT; es = 4; wpe = 1
  case (2); dtype = CUDECOMP_DOUBLE; es = 8; wpe = 1
  case (3); dtype = CUDECOMP_FLOAT_COMPLEX; es = 8; wpe = 2
  case default; dtype = CUDECOMP_DOUBLE_COMPLEX; es = 16; wpe = 2
  end select
  call get_command_argument(0, progname)

  ! one case from the command line, or one per line of a test file
  from_file = .false.
  testfile = ""
  line = ""
  argn = command_argument_count()
  do i = 1, argn
    call get_command_argument(i, arg)
    if ((trim(arg) == "-f" .or. trim(arg) == "--testfile") .and. i < argn) then
      call get_command_argument(i + 1, testfile)
      from_file = .true.
    end if
    line = trim(line)//" "//trim(arg)
  end do
  if (from_file) then
    ncases = 0
    open (newunit=u, file=trim(testfile), status="old", action="read", iostat=stat)
    if (stat /= 0) error stop "cannot open the test file"
    do
      read (u, '(a)', iostat=stat) arg
      if (stat /= 0) exit
      if (len_trim(arg) > 0) ncases = ncases + 1
    end do
    rewind (u)
    allocate (cases(ncases))
    i = 0
    do
      read (u, '(a)', iostat=stat) arg
      if (stat /= 0) exit
      if (len_trim(arg) > 0) then
        i = i + 1
        cases(i) = arg
      end if
    end do
    close (u)
  else
    ncases = 1
    allocate (cases(1))
    cases(1) = line
  end if

  call hipcheck(hipGetDeviceCount(ndev), "hipGetDeviceCount")
  call hipcheck(hipSetDevice(mod(env_int("LOCAL_RANK", rank), ndev)), "hipSetDevice")
  call check(cudecompInit(handle, WORLD_COMM), "cudecompInit")

  nfailed = 0
  call system_clock(c0, rate)
  if (from_file .and. rank == 0) write (*, '(a,i0,a)') "Running ", ncases, " tests..."
  do i = 1, ncases
    if (from_file .and. rank == 0) write (*, '(a,a,a,a)') "command: ", trim(progname), " ", trim(cases(i))
    nfail = 0
    call run_case(trim(cases(i)))
    res = reduce_verdict(min(nfail, 1), i)
    if (rank == 0) then
      if (from_file) then
        if (res /= 0) then
          write (*, '(a)') " FAILED"
        else
          write (*, '(a)') " PASSED"
        end if
      end if
      if (res /= 0) nfailed = nfailed + 1
      if (from_file .and. mod(i, 10) == 0) then
        call system_clock(c1)
        write (*, '(a,i0,a,i0,a,f0.3,a)') "Completed ", i, "/", ncases, " tests, running time ", real(c1 - c0)/real(rate), " s"
      end if
    else if (nfail /= 0) then
      nfailed = nfailed + 1
    end if
  end do
  call check(cudecompFinalize(handle), "cudecompFinalize")
  if (rank == 0) then
    call system_clock(c1)
    if (from_file) write (*, '(a,f0.3,a)') "Completed all tests, running time ", real(c1 - c0)/real(rate), " s,"
    if (nfailed == 0) then
      if (from_file) then
        write (*, '(a)') "Passed all tests."
      else
        write (*, '(a)') "PASSED"
      end if
    else
      write (*, '(a,i0,a,i0,a)') "Failed ", nfailed, "/", ncases, " tests."
    end if
  end if
  if (nfailed /= 0) error stop 1

contains

  subroutine run_case(cmd)
    character(len=*), intent(in) :: cmd
    type(cmdline) :: c
    type(cudecompGridDesc) :: grid_desc
    type(cudecompGridDescConfig) :: config
    type(cudecompGridDescAutotuneOptions) :: options
    type(cudecompPencilInfo) :: p(3)
    integer :: gd(3), gdd(3), pd(2), backend, ac(3), halo(3, 3), pad(3, 3), mo(9), rank_order, ax
    logical :: oop
    integer(int64) :: ws, nmax
    real(real32), pointer, contiguous :: da(:), db(:), dw(:), din(:), dout(:), dtmp(:)
    integer :: hop, from_ax(4), to_ax(4)
    integer(c_int) :: r

    call tokenize(cmd, c)
    gd = 256
    gd(1) = opt_int(c, "--gx", 256)
    gd(2) = opt_int(c, "--gy", 256)
    gd(3) = opt_int(c, "--gz", 256)
    pd(1) = opt_int(c, "--pr", 0)
    pd(2) = opt_int(c, "--pc", 0)
    rank_order = opt_int(c, "--rank-order", 0)
    backend = opt_int(c, "--backend", 0)
    ac(1) = opt_int(c, "--acx", 0)
    ac(2) = opt_int(c, "--acy", 0)
    ac(3) = opt_int(c, "--acz", 0)
    gdd = 0
    call opt_ints(c, "--gd", gdd)
    halo = 0
    pad = 0
    call opt_ints(c, "--hex", halo(:, 1))
    call opt_ints(c, "--hey", halo(:, 2))
    call opt_ints(c, "--hez", halo(:, 3))
    call opt_ints(c, "--pdx", pad(:, 1))
    call opt_ints(c, "--pdy", pad(:, 2))
    call opt_ints(c, "--pdz", pad(:, 3))
    mo = -1
    call opt_ints(c, "--mem_order", mo)
    oop = find_opt(c, "-o") /= 0 .or. find_opt(c, "--out-of-place") /= 0

    call check(cudecompGridDescConfigSetDefaults(config), "cudecompGridDescConfigSetDefaults")
    config%gdims = gd
    config%gdims_dist = gd - gdd
    config%pdims = pd
    config%rank_order = rank_order
    config%transpose_axis_contiguous = (ac /= 0)
    if (find_opt(c, "--mem_order") /= 0) config%transpose_mem_order = reshape(mo, [3, 3])
    call check(cudecompGridDescAutotuneOptionsSetDefaults(options), "cudecompGridDescAutotuneOptionsSetDefaults")
    options%dtype = dtype
    options%transpose_use_inplace_buffers = .not. oop
    if (backend /= 0) then
      config%transpose_comm_backend = backend
    else
      options%autotune_transpose_backend = .true.
    end if
    r = cudecompGridDescCreate(handle, grid_desc, config, options)
    if (r /= CUDECOMP_RESULT_SUCCESS) then
      write (error_unit, '(a,i0)') "cudecompGridDescCreate returned ", r
      nfail = nfail + 1
      return
    end if
    if (.not. from_file .and. rank == 0) &
      write (*, '(a,i0,a,i0,a,a,a)') "running on ", config%pdims(1), " x ", config%pdims(2), " process grid, ", &
      cudecompTransposeCommBackendToString(config%transpose_comm_backend), " transpose backend..."

    do ax = 1, 3
      call check(cudecompGetPencilInfo(handle, grid_desc, p(ax), ax, halo(:, ax), pad(:, ax)), "cudecompGetPencilInfo")
    end do
    call check(cudecompGetTransposeWorkspaceSize(handle, grid_desc, ws), "cudecompGetTransposeWorkspaceSize")
    nmax = max(p(1)%size, p(2)%size, p(3)%size)
    call check(cudecompMalloc(handle, grid_desc, dw, max(ws, 1_int64)*es/4), "cudecompMalloc work")
    call check(cudecompMalloc(handle, grid_desc, da, nmax*es/4), "cudecompMalloc a")
    if (oop) then
      call check(cudecompMalloc(handle, grid_desc, db, nmax*es/4), "cudecompMalloc b")
    else
      db => da
    end if

    call upload(da, p(1), gd)
    from_ax = [1, 2, 3, 2]
    to_ax = [2, 3, 2, 1]
    din => da
    dout => db
    do hop = 1, 4
      select case (hop)
      case (1)
        r = cudecompTransposeXToY(handle, grid_desc, din, dout, dw, dtype, halo(:, 1), halo(:, 2), pad(:, 1), pad(:, 2))
      case (2)
        r = cudecompTransposeYToZ(handle, grid_desc, din, dout, dw, dtype, halo(:, 2), halo(:, 3), pad(:, 2), pad(:, 3))
      case (3)
        r = cudecompTransposeZToY(handle, grid_desc, din, dout, dw, dtype, halo(:, 3), halo(:, 2), pad(:, 3), pad(:, 2))
      case (4)
        r = cudecompTransposeYToX(handle, grid_desc, din, dout, dw, dtype, halo(:, 2), halo(:, 1), pad(:, 2), pad(:, 1))
      end select
      if (r /= CUDECOMP_RESULT_SUCCESS) then
        write (error_unit, '(a,i0,a,i0)') "transpose hop ", hop, " returned ", r
        nfail = nfail + 1
        exit
      end if
      call hipcheck(hipDeviceSynchronize(), "sync")
      call verify(dout, p(to_ax(hop)), gd, hop)
      if (oop) then
        dtmp => din
        din => dout
        dout => dtmp
      end if
    end do

    if (oop) call check(cudecompFree(handle, grid_desc, db), "cudecompFree b")
    call check(cudecompFree(handle, grid_desc, da), "cudecompFree a")
    call check(cudecompFree(handle, grid_desc, dw), "cudecompFree work")
    call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")
  end subroutine run_case

  ! host staging: element e holds value v as (v) or (v, -v) in the element's own word type
  subroutine upload(dev, p, gd)
    real(real32), pointer, contiguous :: dev(:)
    type(cudecompPencilInfo), intent(in) :: p
    integer, intent(in) :: gd(3)
    real(real64), allocatable :: ref(:)
    real(real32), allocatable, target :: h4(:)
    real(real64), allocatable, target :: h8(:)
    integer(int64) :: e
    allocate (ref(p%size))
    call fill_expected(p, gd, ref, -1.0_real64)
    if (es/wpe == 4) then
      allocate (h4(p%size*wpe))
      do e = 1, p%size
        h4((e - 1)*wpe + 1) = real(ref(e), real32)
        if (wpe == 2) h4(e*2) = -real(ref(e), real32)
      end do
      call hipcheck(hipMemcpy(c_loc(dev), c_loc(h4), int(p%size*es, c_size_t), hipMemcpyHostToDevice), "H2D")
    else
      allocate (h8(p%size*wpe))
      do e = 1, p%size
        h8((e - 1)*wpe + 1) = ref(e)
        if (wpe == 2) h8(e*2) = -ref(e)
      end do
      call hipcheck(hipMemcpy(c_loc(dev), c_loc(h8), int(p%size*es, c_size_t), hipMemcpyHostToDevice), "H2D")
    end if
  end subroutine upload

  ! interior cells must equal the closed form exactly (halo cells are unspecified after a transpose)
  subroutine verify(dev, p, gd, hop)
    real(real32), pointer, contiguous :: dev(:)
    type(cudecompPencilInfo), intent(in) :: p
    integer, intent(in) :: gd(3), hop
    real(real64), allocatable :: ref(:)
    real(real32), allocatable, target :: h4(:)
    real(real64), allocatable, target :: h8(:)
    integer(int64) :: e, bad
    real(real64) :: re, im
    allocate (ref(p%size))
    call fill_expected(p, gd, ref, -7.0_real64)
    if (es/wpe == 4) then
      allocate (h4(p%size*wpe))
      call hipcheck(hipMemcpy(c_loc(h4), c_loc(dev), int(p%size*es, c_size_t), hipMemcpyDeviceToHost), "D2H")
    else
      allocate (h8(p%size*wpe))
      call hipcheck(hipMemcpy(c_loc(h8), c_loc(dev), int(p%size*es, c_size_t), hipMemcpyDeviceToHost), "D2H")
    end if
    bad = 0
    do e = 1, p%size
      if (ref(e) == -7.0_real64) cycle
      if (es/wpe == 4) then
        re = h4((e - 1)*wpe + 1)
        im = -re
        if (wpe == 2) im = h4(e*2)
        if (re /= real(real(ref(e), real32), real64) .or. im /= -re) bad = bad + 1
      else
        re = h8((e - 1)*wpe + 1)
        im = -re
        if (wpe == 2) im = h8(e*2)
        if (re /= ref(e) .or. im /= -ref(e)) bad = bad + 1
      end if
    end do
    if (bad /= 0) then
      nfail = nfail + 1
      write (error_unit, '(a,i0,a,i0,a,i0)') "MISMATCH after hop ", hop, ": ", bad, " cells on rank ", rank
    end if
  end subroutine verify

end program transpose_test
