! halo_test.f90 -- Fortran twin of the halo parity test (reference tests/fortran/halo_test.f90,
! tests/ctest/fortran_halo_tests.f90): fill the interior of one pencil with global indices and everything else
! with -1, update the halos of all three dimensions through the Fortran module (one-based `dim`), compare the
! whole pencil -- halos, corners and padding -- with the closed form.
!
! usage: halo_test gx gy gz prow pcol backend axis hx hy hz perx pery perz padx pady padz acflag
program halo_test
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real64
  use cudecomp
  use test_support
  implicit none

  type(cudecompHandle) :: handle
  type(cudecompGridDesc) :: grid_desc
  type(cudecompGridDescConfig) :: config
  type(cudecompPencilInfo) :: p
  integer :: rank, gd(3), pd(2), backend, axis, halo(3), iper(3), pad(3), acflag, ndev, i, dim
  logical :: periods(3)
  integer(int64) :: ws, e, bad
  real(real64), pointer, contiguous :: dbuf(:), dwork(:)
  real(real64), allocatable, target :: host(:)
  real(real64), allocatable :: ref(:)
  integer(c_int) :: res

  gd = [12, 10, 14]
  pd = [1, 1]
  backend = CUDECOMP_HALO_COMM_NCCL
  axis = 1
  halo = [1, 1, 1]
  iper = 1
  pad = 0
  acflag = 0
  do i = 1, 3
    call arg_int(i, gd(i))
    call arg_int(7 + i, halo(i))
    call arg_int(10 + i, iper(i))
    call arg_int(13 + i, pad(i))
  end do
  call arg_int(4, pd(1))
  call arg_int(5, pd(2))
  call arg_int(6, backend)
  call arg_int(7, axis)
  call arg_int(17, acflag)
  periods = (iper /= 0)
  rank = env_int("RANK", 0)

  call hipcheck(hipGetDeviceCount(ndev), "hipGetDeviceCount")
  call hipcheck(hipSetDevice(mod(env_int("LOCAL_RANK", rank), ndev)), "hipSetDevice")
  call check(cudecompInit(handle, WORLD_COMM), "cudecompInit")
  call check(cudecompGridDescConfigSetDefaults(config), "cudecompGridDescConfigSetDefaults")
  config%gdims = gd
  config%pdims = pd
  config%halo_comm_backend = backend
  config%transpose_axis_contiguous = (acflag /= 0)
  call check(cudecompGridDescCreate(handle, grid_desc, config), "cudecompGridDescCreate")
  call check(cudecompGetPencilInfo(handle, grid_desc, p, axis, halo, pad), "cudecompGetPencilInfo")
  call check(cudecompGetHaloWorkspaceSize(handle, grid_desc, axis, halo, ws), "cudecompGetHaloWorkspaceSize")

  call check(cudecompMalloc(handle, grid_desc, dbuf, p%size), "cudecompMalloc data")
  call check(cudecompMalloc(handle, grid_desc, dwork, max(ws, 1_int64)), "cudecompMalloc work")

  allocate (host(p%size), ref(p%size))
  call fill_expected(p, gd, host, -1.0_real64)
  call hipcheck(hipMemcpy(c_loc(dbuf), c_loc(host), int(p%size*8, c_size_t), hipMemcpyHostToDevice), "H2D")

  do dim = 1, 3
    select case (axis)
    case (1)
      res = cudecompUpdateHalosX(handle, grid_desc, dbuf, dwork, CUDECOMP_DOUBLE, halo, periods, dim, pad)
    case (2)
      res = cudecompUpdateHalosY(handle, grid_desc, dbuf, dwork, CUDECOMP_DOUBLE, halo, periods, dim, pad)
    case default
      res = cudecompUpdateHalosZ(handle, grid_desc, dbuf, dwork, CUDECOMP_DOUBLE, halo, periods, dim, pad)
    end select
    call check(res, "cudecompUpdateHalos")
  end do
  call hipcheck(hipDeviceSynchronize(), "sync")
  call hipcheck(hipMemcpy(c_loc(host), c_loc(dbuf), int(p%size*8, c_size_t), hipMemcpyDeviceToHost), "D2H")

  call fill_expected_halo(p, gd, periods, ref)
  bad = 0
  do e = 1, p%size
    if (host(e) /= ref(e)) bad = bad + 1
  end do
  if (bad /= 0) then
    nfail = nfail + 1
    write (*, '(a,i0,a,i0)') "MISMATCH: ", bad, " cells on rank ", rank
  end if

  call check(cudecompFree(handle, grid_desc, dbuf), "cudecompFree data")
  call check(cudecompFree(handle, grid_desc, dwork), "cudecompFree work")
  call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")
  call check(cudecompFinalize(handle), "cudecompFinalize")
  if (nfail /= 0) error stop 2
  write (*, '(a,1x,i0)') "PASS", rank
end program halo_test
