! halo_test.f90 -- Fortran twin of tests/native/halo_test.cpp: fill the interior of one pencil with global indices and
! everything else with -1, update the halos of all three dimensions through the Fortran module (one-based `dim`), compare
! the whole pencil -- halos, corners and padding -- with the closed form.  Command line, test-file mode and output
! protocol of the reference's tests/fortran/halo_test.f90 in FORTRAN conventions (--ax 1..3, --mem_order entries 1..3), so
! the `halo_test*_fortran` configurations of its tests/test_config.yaml drive it unchanged.  Data type from the
! executable's name (halo_test_R32 / _R64 / _C32 / _C64 link to this program; R64 otherwise).
!
!   --gx --gy --gz N   --pr --pc N (0 0 = autotune)   --rank-order 0|1|2   --backend B (0 = autotune)   --ac 0|1
!   --gd a b c   --hex --hey --hez N   --hpx --hpy --hpz 0|1   --pdx --pdy --pdz N   --ax 1|2|3   --mem_order a b c
!   -m accepted, ignored   -f|--testfile FILE
program halo_test
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
    type(cudecompPencilInfo) :: p
    integer :: gd(3), gdd(3), pd(2), backend, axis, halo(3), iper(3), pad(3), mo(3), ac, rank_order, dim, ax2
    logical :: periods(3)
    integer(int64) :: ws, e, bad
    real(real32), pointer, contiguous :: dbuf(:), dwork(:)
    real(real64), allocatable :: init(:), ref(:)
    real(real32), allocatable, target :: h4(:)
    real(real64), allocatable, target :: h8(:)
    real(real64) :: re, im
    integer(c_int) :: r

    call tokenize(cmd, c)
    gd(1) = opt_int(c, "--gx", 256)
    gd(2) = opt_int(c, "--gy", 256)
    gd(3) = opt_int(c, "--gz", 256)
    pd(1) = opt_int(c, "--pr", 0)
    pd(2) = opt_int(c, "--pc", 0)
    rank_order = opt_int(c, "--rank-order", 0)
    backend = opt_int(c, "--backend", 0)
    ac = opt_int(c, "--ac", 0)
    gdd = 0
    call opt_ints(c, "--gd", gdd)
    halo(1) = opt_int(c, "--hex", 1)
    halo(2) = opt_int(c, "--hey", 1)
    halo(3) = opt_int(c, "--hez", 1)
    iper(1) = opt_int(c, "--hpx", 1)
    iper(2) = opt_int(c, "--hpy", 1)
    iper(3) = opt_int(c, "--hpz", 1)
    pad(1) = opt_int(c, "--pdx", 0)
    pad(2) = opt_int(c, "--pdy", 0)
    pad(3) = opt_int(c, "--pdz", 0)
    axis = opt_int(c, "--ax", 1)
    periods = (iper /= 0)
    mo = -1
    call opt_ints(c, "--mem_order", mo)

    call check(cudecompGridDescConfigSetDefaults(config), "cudecompGridDescConfigSetDefaults")
    config%gdims = gd
    config%gdims_dist = gd - gdd
    config%pdims = pd
    config%rank_order = rank_order
    config%transpose_axis_contiguous = (ac /= 0)
    if (find_opt(c, "--mem_order") /= 0) then
      do ax2 = 1, 3
        config%transpose_mem_order(:, ax2) = mo
      end do
    end if
    call check(cudecompGridDescAutotuneOptionsSetDefaults(options), "cudecompGridDescAutotuneOptionsSetDefaults")
    options%dtype = dtype
    options%grid_mode = CUDECOMP_AUTOTUNE_GRID_HALO
    options%halo_axis = axis
    options%halo_extents = halo
    options%halo_periods = periods
    options%halo_padding = pad
    if (backend /= 0) then
      config%halo_comm_backend = backend
    else
      options%autotune_halo_backend = .true.
    end if
    r = cudecompGridDescCreate(handle, grid_desc, config, options)
    if (r /= CUDECOMP_RESULT_SUCCESS) then
      write (error_unit, '(a,i0)') "cudecompGridDescCreate returned ", r
      nfail = nfail + 1
      return
    end if
    if (.not. from_file .and. rank == 0) &
      write (*, '(a,i0,a,i0,a,a,a)') "running on ", config%pdims(1), " x ", config%pdims(2), " process grid, ", &
      cudecompHaloCommBackendToString(config%halo_comm_backend), " halo backend..."

    call check(cudecompGetPencilInfo(handle, grid_desc, p, axis, halo, pad), "cudecompGetPencilInfo")
    call check(cudecompGetHaloWorkspaceSize(handle, grid_desc, axis, halo, ws), "cudecompGetHaloWorkspaceSize")
    call check(cudecompMalloc(handle, grid_desc, dbuf, p%size*es/4), "cudecompMalloc data")
    call check(cudecompMalloc(handle, grid_desc, dwork, max(ws, 1_int64)*es/4), "cudecompMalloc work")

    allocate (init(p%size), ref(p%size))
    call fill_expected(p, gd, init, -1.0_real64)
    call fill_expected_halo(p, gd, periods, ref)
    if (es/wpe == 4) then
      allocate (h4(p%size*wpe))
      do e = 1, p%size
        h4((e - 1)*wpe + 1) = real(init(e), real32)
        if (wpe == 2) h4(e*2) = -real(init(e), real32)
      end do
      call hipcheck(hipMemcpy(c_loc(dbuf), c_loc(h4), int(p%size*es, c_size_t), hipMemcpyHostToDevice), "H2D")
    else
      allocate (h8(p%size*wpe))
      do e = 1, p%size
        h8((e - 1)*wpe + 1) = init(e)
        if (wpe == 2) h8(e*2) = -init(e)
      end do
      call hipcheck(hipMemcpy(c_loc(dbuf), c_loc(h8), int(p%size*es, c_size_t), hipMemcpyHostToDevice), "H2D")
    end if

    do dim = 1, 3
      select case (axis)
      case (1)
        r = cudecompUpdateHalosX(handle, grid_desc, dbuf, dwork, dtype, halo, periods, dim, pad)
      case (2)
        r = cudecompUpdateHalosY(handle, grid_desc, dbuf, dwork, dtype, halo, periods, dim, pad)
      case default
        r = cudecompUpdateHalosZ(handle, grid_desc, dbuf, dwork, dtype, halo, periods, dim, pad)
      end select
      if (r /= CUDECOMP_RESULT_SUCCESS) then
        write (error_unit, '(a,i0,a,i0)') "halo update of dim ", dim, " returned ", r
        nfail = nfail + 1
      end if
    end do
    call hipcheck(hipDeviceSynchronize(), "sync")
    bad = 0
    if (es/wpe == 4) then
      call hipcheck(hipMemcpy(c_loc(h4), c_loc(dbuf), int(p%size*es, c_size_t), hipMemcpyDeviceToHost), "D2H")
      do e = 1, p%size
        re = h4((e - 1)*wpe + 1)
        im = -re
        if (wpe == 2) im = h4(e*2)
        if (re /= real(real(ref(e), real32), real64) .or. im /= -re) bad = bad + 1
      end do
    else
      call hipcheck(hipMemcpy(c_loc(h8), c_loc(dbuf), int(p%size*es, c_size_t), hipMemcpyDeviceToHost), "D2H")
      do e = 1, p%size
        re = h8((e - 1)*wpe + 1)
        im = -re
        if (wpe == 2) im = h8(e*2)
        if (re /= ref(e) .or. im /= -ref(e)) bad = bad + 1
      end do
    end if
    if (bad /= 0) then
      nfail = nfail + 1
      write (error_unit, '(a,i0,a,i0)') "MISMATCH: ", bad, " cells on rank ", rank
    end if

    call check(cudecompFree(handle, grid_desc, dbuf), "cudecompFree data")
    call check(cudecompFree(handle, grid_desc, dwork), "cudecompFree work")
    call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")
  end subroutine run_case

end program halo_test
