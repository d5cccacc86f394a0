! api_test.f90 -- Fortran twin of the C-ABI geometry tests (what tests/ctest/fortran_api_test.f90 covers in the
! reference): defaults, config round trip through create, one-based pencil info / shifted ranks, workspace
! sizes, dtype sizes and backend names.  Needs no GPU.  Checks the module's own contracts in place and prints
! the geometry as text records that tests/test_fortran.py compares with the golden vectors and the oracle.
!
! usage: api_test [rank_order [use_gdims_dist]]      (ranks come from the launcher environment)
program api_test
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64
  use cudecomp
  use test_support
  implicit none

  type(cudecompHandle) :: handle
  type(cudecompGridDesc) :: grid_desc
  type(cudecompGridDescConfig) :: config, before, queried, after_create
  type(cudecompGridDescAutotuneOptions) :: options
  type(cudecompPencilInfo) :: pinfo
  integer :: rank, nranks, rank_order, use_dist, axis, dim, disp, iper
  integer, parameter :: halo(3) = [1, 2, 1], pad(3) = [1, 0, 2]
  integer(int64) :: ws, dsize
  integer(c_int32_t) :: shifted
  integer(c_int) :: res

  rank = env_int("RANK", 0)
  nranks = env_int("WORLD_SIZE", 1)
  rank_order = CUDECOMP_RANK_ORDER_DEFAULT
  use_dist = 0
  call arg_int(1, rank_order)
  call arg_int(2, use_dist)

  call check(cudecompInit(handle, WORLD_COMM), "cudecompInit")

  ! ---- defaults ---------------------------------------------------------------------------------------------
  call check(cudecompGridDescConfigSetDefaults(config), "cudecompGridDescConfigSetDefaults")
  call expect(config%rank_order == CUDECOMP_RANK_ORDER_DEFAULT, "default rank order")
  call expect(config%transpose_comm_backend == CUDECOMP_TRANSPOSE_COMM_MPI_P2P, "default transpose backend")
  call expect(config%halo_comm_backend == CUDECOMP_HALO_COMM_MPI, "default halo backend")
  call expect(all(config%pdims == 0) .and. all(config%gdims == 0) .and. all(config%gdims_dist == 0), "default dims")
  call expect(all(config%transpose_mem_order == -1), "default mem order is unset (-1)")
  call expect(.not. any(config%transpose_axis_contiguous), "default axis contiguous")
  call check(cudecompGridDescAutotuneOptionsSetDefaults(options), "cudecompGridDescAutotuneOptionsSetDefaults")
  call expect(options%n_warmup_trials == 3 .and. options%n_trials == 5, "default trial counts")
  call expect(options%grid_mode == CUDECOMP_AUTOTUNE_GRID_TRANSPOSE, "default grid mode")
  call expect(options%dtype == CUDECOMP_DOUBLE, "default autotune dtype")
  call expect(options%halo_axis == 1, "default halo axis is one-based x")
  call expect(all(options%transpose_op_weights == 1.0d0), "default op weights")
  call expect(.not. any(options%transpose_use_inplace_buffers), "default in-place flags")
  call expect(c_sizeof(config) == 104 .and. c_sizeof(options) == 320 .and. c_sizeof(pinfo) == 96, "struct sizes")

  ! ---- create: config and options come back in Fortran conventions ---------------------------------------------------
  config%gdims = [9, 10, 11]
  if (use_dist /= 0) config%gdims_dist = [8, 9, 10]
  if (nranks == 4) then
    config%pdims = [2, 2]
  else
    config%pdims = [1, nranks]
  end if
  config%rank_order = rank_order
  before = config
  options%halo_axis = 3
  res = cudecompGridDescCreate(handle, grid_desc, config, options)
  call check(res, "cudecompGridDescCreate with options")
  ! an unset order comes back as "unset" in one-based terms (C -1 -> 0, as in the reference module) and a
  ! config that went through create can be fed to create again (done below)
  call expect(all(config%transpose_mem_order == 0), "unset mem order is reported as 0 after create")
  after_create = config
  call expect(all(config%gdims == before%gdims) .and. all(config%pdims == before%pdims), "dims kept by create")
  call expect(options%halo_axis == 3, "halo_axis restored after create")
  call check(cudecompGetGridDescConfig(handle, grid_desc, queried), "cudecompGetGridDescConfig")
  call expect(all(queried%gdims == before%gdims) .and. all(queried%pdims == before%pdims), "queried dims")
  call expect(all(queried%gdims_dist == before%gdims_dist), "queried gdims_dist")
  call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")

  ! explicit one-based memory orders round-trip (position, axis): y-pencils stored (y, z, x)
  config = before
  config%transpose_mem_order(:, 1) = [1, 2, 3]
  config%transpose_mem_order(:, 2) = [2, 3, 1]
  config%transpose_mem_order(:, 3) = [3, 1, 2]
  call check(cudecompGridDescCreate(handle, grid_desc, config), "cudecompGridDescCreate with mem order")
  call expect(all(config%transpose_mem_order(:, 2) == [2, 3, 1]), "explicit mem order restored")
  call check(cudecompGetGridDescConfig(handle, grid_desc, queried), "cudecompGetGridDescConfig (mem order)")
  call expect(all(queried%transpose_mem_order == config%transpose_mem_order), "queried mem order is one-based")
  call check(cudecompGetPencilInfo(handle, grid_desc, pinfo, 2), "pencil info with mem order")
  call expect(all(pinfo%order == [2, 3, 1]), "pencil order follows one-based mem order")
  call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy (mem order)")

  ! ---- geometry records ---------------------------------------------------------------------------------------------
  config = after_create
  call check(cudecompGridDescCreate(handle, grid_desc, config), "cudecompGridDescCreate without options")
  do axis = 1, 3
    call check(cudecompGetPencilInfo(handle, grid_desc, pinfo, axis, halo, pad), "cudecompGetPencilInfo")
    write (*, '(a,2(1x,i0),18(1x,i0),1x,i0)') "PINFO", rank, axis, pinfo%shape, pinfo%lo, pinfo%hi, pinfo%order, &
      pinfo%halo_extents, pinfo%padding, pinfo%size
    call check(cudecompGetHaloWorkspaceSize(handle, grid_desc, axis, halo, ws), "cudecompGetHaloWorkspaceSize")
    write (*, '(a,2(1x,i0),1x,i0)') "HALOWS", rank, axis, ws
  end do
  call check(cudecompGetPencilInfo(handle, grid_desc, pinfo, 1), "cudecompGetPencilInfo without optionals")
  call expect(all(pinfo%halo_extents == 0) .and. all(pinfo%padding == 0), "optional halo / padding default to 0")
  call expect(all(pinfo%order == [1, 2, 3]), "default order is one-based")
  call expect(pinfo%lo(1) == 1 .and. pinfo%hi(1) == 9, "x range of an x-pencil is one-based")
  call check(cudecompGetTransposeWorkspaceSize(handle, grid_desc, ws), "cudecompGetTransposeWorkspaceSize")
  write (*, '(a,1x,i0,1x,i0)') "TRANSWS", rank, ws
  do dim = 1, 3
    do disp = -1, 1, 2
      do iper = 0, 1
        call check(cudecompGetShiftedRank(handle, grid_desc, 1, dim, disp, iper == 1, shifted), &
                   "cudecompGetShiftedRank")
        write (*, '(a,5(1x,i0))') "SHIFT", rank, dim, disp, iper, shifted
      end do
    end do
  end do
  ! out-of-range one-based arguments are rejected, not wrapped
  res = cudecompGetPencilInfo(handle, grid_desc, pinfo, 0)
  call expect(res == CUDECOMP_RESULT_INVALID_USAGE, "axis 0 is invalid in the one-based API")
  res = cudecompGetPencilInfo(handle, grid_desc, pinfo, 4)
  call expect(res == CUDECOMP_RESULT_INVALID_USAGE, "axis 4 is invalid")
  res = cudecompGetShiftedRank(handle, grid_desc, 1, 0, 1, .false., shifted)
  call expect(res == CUDECOMP_RESULT_INVALID_USAGE, "dim 0 is invalid")
  call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")

  ! ---- scalars ---------------------------------------------------------------------------------------------
  call check(cudecompGetDataTypeSize(CUDECOMP_FLOAT, dsize), "dtype size")
  call expect(dsize == 4, "float size")
  call check(cudecompGetDataTypeSize(CUDECOMP_DOUBLE, dsize), "dtype size")
  call expect(dsize == 8, "double size")
  call check(cudecompGetDataTypeSize(CUDECOMP_FLOAT_COMPLEX, dsize), "dtype size")
  call expect(dsize == 8, "float complex size")
  call check(cudecompGetDataTypeSize(CUDECOMP_DOUBLE_COMPLEX, dsize), "dtype size")
  call expect(dsize == 16, "double complex size")
  res = cudecompGetDataTypeSize(0, dsize)
  call expect(res == CUDECOMP_RESULT_INVALID_USAGE, "invalid dtype")
  call expect(cudecompTransposeCommBackendToString(CUDECOMP_TRANSPOSE_COMM_NCCL) == "NCCL", "NCCL name")
  call expect(cudecompTransposeCommBackendToString(CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL) == "MPI_P2P (pipelined)", &
              "MPI_P2P_PL name")
  call expect(cudecompTransposeCommBackendToString(99) == "ERROR", "bad transpose backend name")
  call expect(cudecompHaloCommBackendToString(CUDECOMP_HALO_COMM_MPI_BLOCKING) == "MPI (blocking)", "halo name")
  call expect(cudecompHaloCommBackendToString(-3) == "ERROR", "bad halo backend name")

  call check(cudecompFinalize(handle), "cudecompFinalize")
  if (nfail /= 0) error stop 2
  write (*, '(a,1x,i0)') "DONE", rank
end program api_test
