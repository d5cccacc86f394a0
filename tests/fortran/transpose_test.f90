! transpose_test.f90 -- Fortran twin of the transpose parity test (reference tests/fortran/transpose_test.f90 and
! tests/ctest/fortran_transpose_tests.f90 cover the same ground): X->Y->Z->Y->X through the Fortran module on
! device buffers, every stage compared element-for-element with the closed-form pencil contents.
!
! usage: transpose_test gx gy gz prow pcol backend acx acy acz hx hy hz inplace dtype
!   dtype: 1 real32, 2 real64, 3 complex32, 4 complex64; ranks come from the launcher environment
program transpose_test
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64, real32, real64
  use cudecomp
  use test_support
  implicit none

  type(cudecompHandle) :: handle
  type(cudecompGridDesc) :: grid_desc
  type(cudecompGridDescConfig) :: config
  type(cudecompPencilInfo) :: px, py, pz
  integer :: rank, nranks, gd(3), pd(2), backend, ac(3), halo(3), inplace, dtype_sel, ndev, dtype, wpe, i
  integer(int64) :: ws, nmax, es
  integer(cudecomp_stream_kind) :: stream
  ! device memory, handed out as 4-byte words whatever the element type is (the API ignores type and rank)
  real(real32), pointer, contiguous :: dbuf_a(:), dbuf_b(:), dbuf_c(:), dwork(:)
  real(real64), pointer, contiguous :: typed_r8(:)
  complex(real32), pointer, contiguous :: typed_c4(:)
  complex(real64), pointer, contiguous :: typed_c8(:)
  type(c_ptr) :: raw

  gd = [16, 12, 10]
  pd = [1, 1]
  backend = CUDECOMP_TRANSPOSE_COMM_NCCL
  ac = 0
  halo = 0
  inplace = 0
  dtype_sel = 2
  do i = 1, 3
    call arg_int(i, gd(i))
    call arg_int(6 + i, ac(i))
    call arg_int(9 + i, halo(i))
  end do
  call arg_int(4, pd(1))
  call arg_int(5, pd(2))
  call arg_int(6, backend)
  call arg_int(13, inplace)
  call arg_int(14, dtype_sel)
  rank = env_int("RANK", 0)
  nranks = env_int("WORLD_SIZE", 1)
  select case (dtype_sel)
  case (1); dtype = CUDECOMP_FLOAT; es = 4; wpe = 1
  case (2); dtype = CUDECOMP_DOUBLE; es = 8; wpe = 1
  case (3); dtype = CUDECOMP_FLOAT_COMPLEX; es = 8; wpe = 2
  case default; dtype = CUDECOMP_DOUBLE_COMPLEX; es = 16; wpe = 2
  end select

  call hipcheck(hipGetDeviceCount(ndev), "hipGetDeviceCount")
  call hipcheck(hipSetDevice(mod(env_int("LOCAL_RANK", rank), ndev)), "hipSetDevice")
  call check(cudecompInit(handle, WORLD_COMM), "cudecompInit")
  call check(cudecompGridDescConfigSetDefaults(config), "cudecompGridDescConfigSetDefaults")
  config%gdims = gd
  config%pdims = pd
  config%transpose_comm_backend = backend
  config%transpose_axis_contiguous = (ac /= 0)
  call check(cudecompGridDescCreate(handle, grid_desc, config), "cudecompGridDescCreate")

  call check(cudecompGetPencilInfo(handle, grid_desc, px, 1, halo), "pencil info x")
  call check(cudecompGetPencilInfo(handle, grid_desc, py, 2, halo), "pencil info y")
  call check(cudecompGetPencilInfo(handle, grid_desc, pz, 3, halo), "pencil info z")
  call check(cudecompGetTransposeWorkspaceSize(handle, grid_desc, ws), "cudecompGetTransposeWorkspaceSize")
  nmax = max(px%size, py%size, pz%size)

  ! typed allocation entry points (sizes in elements); the test itself addresses memory as words
  call check(cudecompMalloc(handle, grid_desc, typed_r8, 16_int64), "cudecompMalloc real64")
  call check(cudecompFree(handle, grid_desc, typed_r8), "cudecompFree real64")
  call check(cudecompMalloc(handle, grid_desc, typed_c4, 16_int64), "cudecompMalloc complex32")
  call check(cudecompFree(handle, grid_desc, typed_c4), "cudecompFree complex32")
  call check(cudecompMalloc(handle, grid_desc, typed_c8, 16_int64), "cudecompMalloc complex64")
  call check(cudecompFree(handle, grid_desc, typed_c8), "cudecompFree complex64")
  call check(cudecompMalloc(handle, grid_desc, raw, 256_int64), "cudecompMalloc c_ptr")
  call expect(c_associated(raw), "raw allocation is non-null")
  call check(cudecompFree(handle, grid_desc, raw), "cudecompFree c_ptr")
  call expect(.not. c_associated(raw), "raw pointer reset by free")

  call check(cudecompMalloc(handle, grid_desc, dwork, max(ws, 1_int64)*es/4), "cudecompMalloc work")
  call check(cudecompMalloc(handle, grid_desc, dbuf_a, nmax*es/4), "cudecompMalloc a")
  if (inplace /= 0) then
    dbuf_b => dbuf_a
    dbuf_c => dbuf_a
  else
    call check(cudecompMalloc(handle, grid_desc, dbuf_b, nmax*es/4), "cudecompMalloc b")
    call check(cudecompMalloc(handle, grid_desc, dbuf_c, nmax*es/4), "cudecompMalloc c")
  end if
  call hipcheck(hipStreamCreate(stream), "hipStreamCreate")

  ! X pencil <- closed form; then walk the cycle.  The first op uses the null stream through the optional
  ! argument's default, the others an explicit stream.
  call upload(dbuf_a, px, -1.0_real64)
  call check(cudecompTransposeXToY(handle, grid_desc, dbuf_a, dbuf_b, dwork, dtype, halo, halo), "XToY")
  call hipcheck(hipDeviceSynchronize(), "sync")
  call verify(dbuf_b, py, "XToY")
  call check(cudecompTransposeYToZ(handle, grid_desc, dbuf_b, dbuf_c, dwork, dtype, halo, halo, stream=stream), "YToZ")
  call hipcheck(hipStreamSynchronize(stream), "sync")
  call verify(dbuf_c, pz, "YToZ")
  call check(cudecompTransposeZToY(handle, grid_desc, dbuf_c, dbuf_b, dwork, dtype, input_halo_extents=halo, &
                                   output_halo_extents=halo, stream=stream), "ZToY")
  call hipcheck(hipStreamSynchronize(stream), "sync")
  call verify(dbuf_b, py, "ZToY")
  call check(cudecompTransposeYToX(handle, grid_desc, dbuf_b, dbuf_a, dwork, dtype, halo, halo, stream=stream), "YToX")
  call hipcheck(hipStreamSynchronize(stream), "sync")
  call verify(dbuf_a, px, "YToX")

  if (inplace == 0) then
    call check(cudecompFree(handle, grid_desc, dbuf_b), "cudecompFree b")
    call check(cudecompFree(handle, grid_desc, dbuf_c), "cudecompFree c")
  end if
  call check(cudecompFree(handle, grid_desc, dbuf_a), "cudecompFree a")
  call check(cudecompFree(handle, grid_desc, dwork), "cudecompFree work")
  call check(cudecompGridDescDestroy(handle, grid_desc), "cudecompGridDescDestroy")
  call check(cudecompFinalize(handle), "cudecompFinalize")
  if (nfail /= 0) error stop 2
  write (*, '(a,1x,i0)') "PASS", rank

contains

  ! host staging: element e holds value v as (v) or (v, -v) in the element's own word type
  subroutine upload(dev, p, outside)
    real(real32), pointer, contiguous :: dev(:)
    type(cudecompPencilInfo), intent(in) :: p
    real(real64), intent(in) :: outside
    real(real64), allocatable :: ref(:)
    real(real32), allocatable, target :: h4(:)
    real(real64), allocatable, target :: h8(:)
    integer(int64) :: e
    allocate (ref(p%size))
    call fill_expected(p, gd, ref, outside)
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
  subroutine verify(dev, p, what)
    real(real32), pointer, contiguous :: dev(:)
    type(cudecompPencilInfo), intent(in) :: p
    character(len=*), intent(in) :: what
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
      else
        re = h8((e - 1)*wpe + 1)
        im = -re
        if (wpe == 2) im = h8(e*2)
      end if
      if (re /= ref(e) .or. im /= -ref(e)) bad = bad + 1
    end do
    if (bad /= 0) then
      nfail = nfail + 1
      write (*, '(a,a,a,i0,a,i0)') "MISMATCH after ", what, ": ", bad, " cells on rank ", rank
    end if
  end subroutine verify

end program transpose_test
