# Julia-side binding of libgsr_hip.so for GaussianSplatting.jl (reference @ v2.0.0).
#
# Could not be executed in the build environment (no Julia there); kept small and field-for-field
# identical to the ctypes binding that IS tested (gaussiansplatting.jl_amd/_lib.py + rasterizer.py).
#
# Design (what makes it a drop-in — every caller of the reference runs unchanged):
#   * callers keep holding the reference's own `GaussianRasterizer` — `Trainer{R <: GaussianRasterizer}`
#     (training.jl:185-194,225), the functor `rast(points, opacities, scales, rotations, f_dc, f_rest; ...)`
#     (rasterizer.jl:200-253, called at training.jl:646 / :501) and `rast.gstate.radii` / `rast.gstate.∇means_2d`
#     (strategy.jl:85-86) all keep working because nothing about that object changes;
#   * the native handle lives in a side table keyed by the rasterizer (`enable_hip_native!(rast)`); rasterizers
#     that were not enabled — e.g. the sky dome's second `GaussianRasterizer` (sky_dome.jl:144,191) unless it is
#     enabled too — fall through to the reference kernels via `invoke`;
#   * NO dispatch on keyword arguments (Julia has none): the two methods below are more specific than the
#     reference's only in their POSITIONAL types (ROCArray instead of AbstractArray), and branch on the side table;
#   * `rasterize` and `∇rasterize` are overridden; the reference's `ChainRulesCore.rrule` (rasterizer.jl:552-573) calls
#     exactly these two generics, so Zygote and the functor prologue are the reference's own code.  ONE rrule method is added
#     (same body as the reference's, ROCArray positional types): it tells the forward that a pullback will follow.  A
#     `rasterize` that is called directly — the reference's non-AD branch (rasterizer.jl:214-248): `validate`
#     (training.jl:501-504), the GUI (gui/worker.jl:654-657), scripts/render-views.jl — passes GSR_FORWARD_ONLY: same image,
#     no backward state written (a third of the forward's HBM traffic).  `enable_hip_native!(rast;
#     forward_only_outside_ad=false)` keeps the state for callers that run `∇rasterize` by hand after a bare `rasterize`;
#   * the library writes `radii` and `∇means_2d` straight into `rast.gstate` (gsr_aux.radii / gsr_grads.vmeans2d) and
#     the image into `rast.image`, so the reference's state object stays truthful.
#
# Replaces: src/rasterization/rasterizer.jl:255-408 (rasterize), :416-550 (∇rasterize).
module GaussianSplattingHipNative

using AMDGPU, StaticArrays
import KernelAbstractions as KA
import GPUArrays
import GaussianSplatting
import GaussianSplatting: GaussianRasterizer, GeometryState, Camera, resolution, n_color_features, rasterize, ∇rasterize
import ChainRulesCore
import ChainRulesCore: NoTangent, unthunk

const LIB = get(ENV, "GSR_HIP_LIB", "libgsr_hip.so")

struct GsrConfig
    width::Int32; height::Int32; mode::Int32
    near_plane::Float32; far_plane::Float32; radius_clip::Int32; blur_eps::Float32; flags::UInt32
    bins_budget_bytes::UInt64  # 0 = default
    ssim_precision::Int32      # per handle: 0 = process default (ssim_exact!), 1 fast, 2 exact            (ABI 6 encoding:
    preprocess_form::Int32     # per handle: 0 = process default (preprocess_form!), 1 direct, 2 aggregating   0 = default)
    form_tuner::Int32          # ABI 6: 0 default (on), 1 off, 2 on — the handle measures the binning form on 4K-class grids
    grad_precision::Int32      # ABI 6: 0 default, 1 the reference's arithmetic (accurate exp / division per pixel + its fp32 trees for
                               #        ∇scales / ∇rotations), 2 accurate exp / division + the float64 chain
end
struct GsrInputs
    n::Int32; n_coeffs::Int32; sh_degree::Int32
    means::Ptr{Float32}; shs::Ptr{Float32}; opacities::Ptr{Float32}
    scales::Ptr{Float32}; rotations::Ptr{Float32}
    background::NTuple{3, Float32}
end
struct GsrCamera
    R::NTuple{9, Float32}; t::NTuple{3, Float32}; focal::NTuple{2, Float32}
    principal::NTuple{2, Float32}; camera_center::NTuple{3, Float32}
    R_dev::Ptr{Float32}; t_dev::Ptr{Float32}
end
struct GsrAux; covisibilities::Ptr{UInt8}; uncertainties::Ptr{Float32}; radii::Ptr{Int32}; flags::UInt32; reserved::UInt32; end
const GSR_FORWARD_ONLY = 0x00000001  # gsr_aux.flags: this forward will not be differentiated (no backward state kept)
struct GsrStats
    n_rendered::Int64; n_visible::Int32; max_tile_instances::Int32; generation::UInt64
    bins_bytes::Int64; compact_binning::Int32; preprocess_form::Int32
    # ABI 6: the handle's view history (include/gsr_policy.h), cumulative since gsr_create
    bins_regrowths::UInt32; compact_fallbacks::UInt32; tuner_rearms::UInt32; scratch_regrowths::UInt32
    fused_relaunches::UInt32; held_views::UInt32; bin_capacity::UInt32; tuner_form::Int32; tuner_ms::NTuple{2, Float32}
    tier_tiles::NTuple{3, UInt32}; reserved::UInt32
end
struct GsrGrads
    vmeans::Ptr{Float32}; vshs::Ptr{Float32}; vopacities::Ptr{Float32}
    vscales::Ptr{Float32}; vrotations::Ptr{Float32}; vR::Ptr{Float32}; vt::Ptr{Float32}
    vcolors::Ptr{Float32}          # C_NULL: classic form (∇shs written); see gsr.h for the factored multi-view form
    vmeans2d::Ptr{Float32}         # rast.gstate.∇means_2d
    forward_generation::UInt64     # pairs the pullback with its forward (0 = unchecked)
    flags::UInt32; reserved::UInt32 # GSR_GRADS_*
end
struct GsrTailState
    theta::NTuple{6, Ptr{Float32}}; mu::NTuple{6, Ptr{Float32}}; nu::NTuple{6, Ptr{Float32}}
    lr::NTuple{6, Float32}; current_step::NTuple{6, UInt32}
    beta1::Float32; beta2::Float32; eps::Float32; scale_dims::Int32
    shs::Ptr{Float32}; opacities_act::Ptr{Float32}; scales_act::Ptr{Float32}
    vmeans2d::Ptr{Float32}; forward_generation::UInt64
    flags::UInt32; reserved::UInt32
end
const EMPTY_STATS = GsrStats(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, (0f0, 0f0), (0x0, 0x0, 0x0), 0)
const GSR_GRADS_COLOR_COTANGENT = 0x00000001  # channels >= 4 of the cotangent (depth, alpha, normal) are zeros: ONLY valid for the
                                              # buffer loss_l1_ssim! wrote for this forward (the library checks pointer + generation)

check(rc) = rc == 0 || error(unsafe_string(ccall((:gsr_last_error_string, LIB), Cstring, ())))

# GSR_ABI_VERSION of the include/gsr.h these struct definitions mirror; checked (with the seven struct sizes) against
# the loaded library by the first enable_hip_native!: a stale libgsr_hip.so or a stale binding fails here, loudly.
const GSR_ABI_VERSION = 6
const ABI_CHECKED = Ref(false)
function check_abi()
    ABI_CHECKED[] && return
    check(ccall((:gsr_check_abi, LIB), Cint, (Cint, Csize_t, Csize_t, Csize_t, Csize_t, Csize_t, Csize_t, Csize_t), GSR_ABI_VERSION,
        sizeof(GsrConfig), sizeof(GsrInputs), sizeof(GsrCamera), sizeof(GsrAux), sizeof(GsrStats), sizeof(GsrGrads),
        sizeof(GsrTailState)))
    ABI_CHECKED[] = true
end
dptr(::Type{T}, x) where T = x === nothing ? Ptr{T}(C_NULL) : Ptr{T}(UInt(pointer(x)))
dptr(x) = dptr(Float32, x)
hipstream() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))  # the task-local stream (gui/worker.jl:47-51)

# ---- side table: reference rasterizer -> native handle ----
mutable struct NativeState
    handle::Ptr{Cvoid}
    generation::UInt64
    forward_only_outside_ad::Bool  # a bare `rasterize` (no rrule around it) keeps no backward state
    pullback_follows::Bool         # set by the rrule below for the forward it is about to run
    stats::GsrStats                # of the last forward (view history: `history(rast)`)
end
const NATIVE = WeakKeyDict{GaussianRasterizer, NativeState}()
const NATIVE_LOCK = ReentrantLock()
native(rast::GaussianRasterizer) = lock(() -> get(NATIVE, rast, nothing), NATIVE_LOCK)

"""
    enable_hip_native!(rast; reference_tile_lists=false, forward_only_outside_ad=true, ssim_exact=nothing, preprocess_form=nothing,
                       form_tuner=nothing, grad_precision=:default)

Route `rasterize` / `∇rasterize` on this rasterizer through libgsr_hip.so.  Width / height / mode / near / far
are the rasterizer's own (rasterizer.jl:60-90).  `forward_only_outside_ad`: a `rasterize` that is not being
differentiated (no `rrule` around it) is rendered with GSR_FORWARD_ONLY — `∇rasterize` after it is an error.
`ssim_exact` (`nothing` | `false` | `true`) and `preprocess_form` (`nothing` | `0` | `1`) pin the two behaviour switches for THIS
rasterizer (ABI 5; constructor keywords, as the reference's knobs are: rasterizer.jl:60-65); `nothing` follows the process-wide
default (`ssim_exact!`, `preprocess_form!`) — so a GUI render task and a trainer in one process cannot disturb each other.
`form_tuner` (`nothing` | `false` | `true`): whether the handle measures the binning form on 4K-class grids (same outputs either
way).  `grad_precision` (`:default` | `:accurate` | `:fp32_reference`): the backward's arithmetic on needle-shaped splats —
`:accurate` = libm exp + IEEE division per pixel (+12 % of ∇render!; ∇means of a 90 : 1 needle 4e-4 -> 2e-5 from float64),
`:fp32_reference` = that and ∇scales / ∇rotations by the reference's own fp32 expression trees (projection.jl:132-257,
render.jl:302-366) instead of the library's float64 chain — for reference-parity runs.
Returns `rast`.
"""
function enable_hip_native!(rast::GaussianRasterizer; reference_tile_lists::Bool = false, forward_only_outside_ad::Bool = true,
                            ssim_exact::Union{Nothing, Bool} = nothing, preprocess_form::Union{Nothing, Integer} = nothing,
                            form_tuner::Union{Nothing, Bool} = nothing, grad_precision::Symbol = :default)
    native(rast) === nothing || return rast
    check_abi()
    c, w, h = size(rast.image)
    href = Ref{Ptr{Cvoid}}()
    check(ccall((:gsr_create, LIB), Cint, (Ref{GsrConfig}, Ref{Ptr{Cvoid}}),
        GsrConfig(w, h, c, rast.near_plane, rast.far_plane, 3, 0.3f0, reference_tile_lists ? 0x2 : 0x0, 0,
                  # ABI 6: 0 = default, 1 / 2 the explicit choices
                  ssim_exact === nothing ? Int32(0) : Int32(ssim_exact ? 2 : 1),
                  preprocess_form === nothing ? Int32(0) : Int32(preprocess_form != 0 ? 2 : 1),
                  form_tuner === nothing ? Int32(0) : Int32(form_tuner ? 2 : 1),
                  Int32(grad_precision === :fp32_reference ? 1 : grad_precision === :accurate ? 2 : 0)),
        href))
    st = NativeState(href[], 0, forward_only_outside_ad, false, EMPTY_STATS)
    finalizer(s -> ccall((:gsr_destroy, LIB), Cint, (Ptr{Cvoid},), s.handle), st)
    lock(() -> (NATIVE[rast] = st), NATIVE_LOCK)
    return rast
end

"""
    loss_l1_ssim!(rast, target; λ_dssim=0.2f0) -> (loss::ROCVector{Float32} of length 1, vpixels)

The photometric loss head of `step!` (training.jl:656,684-694) fused with its pullback (gsr_loss_l1_ssim) on `rast.image`, the render
of the forward just run: `target` is the (W,H,3) device image.  The returned cotangent has exact zeros above the colour channels and
is the ONLY buffer `backward_trainer_tail!(...; color_cotangent=true)` accepts — the library remembers which buffer its loss head
wrote for which forward, so a cotangent with depth / normal terms can never be mistaken for it (ADVICE r5: the flag used to be a
sticky, unchecked promise).
"""
function loss_l1_ssim!(rast::GaussianRasterizer, target; λ_dssim::Float32 = 0.2f0)
    st = native(rast)
    st === nothing && error("loss_l1_ssim! needs enable_hip_native!(rast)")
    loss = AMDGPU.zeros(Float32, 1); vpixels = similar(rast.image)
    check(ccall((:gsr_loss_l1_ssim, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
        st.handle, dptr(rast.image), dptr(target), λ_dssim, dptr(loss), dptr(vpixels), hipstream()))
    return loss, vpixels
end

"""
    history(rast) -> NamedTuple

The native handle's view history after the last forward (gsr_stats, ABI 6): how often the key bins were regrown, a view fell back
to compact binning, the form tuner started over, a scratch buffer was reallocated — in a steady training run none of them moves.
"""
function history(rast::GaussianRasterizer)
    s = native(rast).stats
    return (; s.bins_regrowths, s.compact_fallbacks, s.tuner_rearms, s.scratch_regrowths, s.fused_relaunches, s.held_views,
            s.bin_capacity, s.tuner_form, s.compact_binning, s.preprocess_form)
end

function disable_hip_native!(rast::GaussianRasterizer)
    lock(() -> delete!(NATIVE, rast), NATIVE_LOCK)  # the finalizer of the NativeState destroys the handle
    return rast
end

function _structs(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, camera::Camera, sh_degree, background)
    K = camera.intrinsics
    R = SMatrix{3, 3, Float32}(camera.w2c[1:3, 1:3]); t = SVector{3, Float32}(camera.w2c[1:3, 4])
    inp = GsrInputs(size(means_3d, 2), size(shs, 2), sh_degree, dptr(means_3d), dptr(shs), dptr(opacities),
        dptr(scales), dptr(rotations), Tuple(background))
    cam = GsrCamera(Tuple(R), Tuple(t), Tuple(K.focal), Tuple(K.principal), Tuple(camera.camera_center),
        dptr(R_w2c), dptr(t_w2c))
    return inp, cam
end

const RM = ROCMatrix{Float32}
const R3 = ROCArray{Float32, 3}
const REF_FWD_SIG = Tuple{AbstractMatrix{Float32}, AbstractArray{Float32, 3}, AbstractMatrix{Float32},
                          AbstractMatrix{Float32}, AbstractMatrix{Float32}, Any, Any}
const REF_BWD_SIG = Tuple{AbstractArray{Float32, 3}, AbstractMatrix{Float32}, AbstractArray{Float32, 3},
                          AbstractMatrix{Float32}, AbstractMatrix{Float32}, AbstractMatrix{Float32},
                          AbstractVector{Int32}, Any, Any}

# rasterize(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c; rast, camera, ...) — rasterizer.jl:255-408
function GaussianSplatting.rasterize(means_3d::RM, shs::R3, opacities::RM, scales::RM, rotations::RM,
        R_w2c = nothing, t_w2c = nothing;
        rast::GaussianRasterizer, camera::Camera, sh_degree::Int, background::SVector{3, Float32},
        covisibilities = nothing, uncertainties = nothing)
    st = native(rast)
    st === nothing && return invoke(rasterize, REF_FWD_SIG, means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c;
        rast, camera, sh_degree, background, covisibilities, uncertainties)
    n = size(means_3d, 2)
    if length(rast.gstate) < n  # rasterizer.jl:275-278: the reference's own grow-only GeometryState
        KA.unsafe_free!(rast.gstate)
        rast.gstate = GPUArrays.@uncached GeometryState(KA.get_backend(rast), n; n_features=n_color_features(rast.mode))
    end
    inp, cam = _structs(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, camera, sh_degree, background)
    keep = st.pullback_follows || !st.forward_only_outside_ad
    st.pullback_follows = false
    aux = GsrAux(dptr(UInt8, covisibilities), dptr(uncertainties), dptr(Int32, rast.gstate.radii),
        keep ? 0x00000000 : GSR_FORWARD_ONLY, 0x00000000)
    stats = Ref(EMPTY_STATS)
    check(ccall((:gsr_forward, LIB), Cint,
        (Ptr{Cvoid}, Ref{GsrInputs}, Ref{GsrCamera}, Ptr{Float32}, Ref{GsrAux}, Ptr{Cvoid}, Ref{GsrStats}),
        st.handle, inp, cam, dptr(rast.image), aux, hipstream(), stats))
    st.generation = stats[].generation
    st.stats = stats[]
    return rast.image  # aliased, overwritten by the next call (rasterizer.jl:407)
end

# ∇rasterize(vpixels, means_3d, shs, scales, rotations, opacities, radii, R_w2c, t_w2c; rast, ...) — rasterizer.jl:416-550
# (called by the reference's rrule pullback, rasterizer.jl:565-570, with radii = rast.gstate.radii)
function GaussianSplatting.∇rasterize(vpixels::R3, means_3d::RM, shs::R3, scales::RM, rotations::RM, opacities::RM,
        radii::ROCVector{Int32}, R_w2c = nothing, t_w2c = nothing;
        rast::GaussianRasterizer, camera::Camera, sh_degree::Int, background::SVector{3, Float32})
    st = native(rast)
    st === nothing && return invoke(∇rasterize, REF_BWD_SIG, vpixels, means_3d, shs, scales, rotations, opacities,
        radii, R_w2c, t_w2c; rast, camera, sh_degree, background)
    # fresh gradient arrays per call, as the reference (rasterizer.jl:437-445); the library overwrites every element
    vmeans = similar(means_3d); vshs = similar(shs); vopac = similar(opacities)
    vscales = similar(scales); vrot = similar(rotations)
    vR = R_w2c === nothing ? nothing : AMDGPU.zeros(Float32, 3, 3)
    vt = R_w2c === nothing ? nothing : AMDGPU.zeros(Float32, 3)
    inp, cam = _structs(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, camera, sh_degree, background)
    g = GsrGrads(dptr(vmeans), dptr(vshs), dptr(vopac), dptr(vscales), dptr(vrot), dptr(vR), dptr(vt),
        Ptr{Float32}(C_NULL), Ptr{Float32}(UInt(pointer(rast.gstate.∇means_2d))), st.generation,
        # (no GSR_GRADS_COLOR_COTANGENT here: under the rrule the cotangent is Zygote's, not the fused loss head's)
        UInt32(0), UInt32(0))
    check(ccall((:gsr_backward, LIB), Cint,
        (Ptr{Cvoid}, Ref{GsrInputs}, Ref{GsrCamera}, Ptr{Float32}, Ref{GsrGrads}, Ptr{Cvoid}),
        st.handle, inp, cam, dptr(vpixels), g, hipstream()))
    return vmeans, vshs, vopac, vscales, vrot, vR, vt
end

# The pullback hands over `unthunk(vpixels)` (rasterizer.jl:567): under Zygote that can be a `FillArrays.Fill`, a
# `Base.ReshapedArray` or a CPU array instead of a ROCArray.  Without this method dispatch would quietly fall through to the
# reference's kernels for an ENABLED rasterizer (correct, but an invisible performance cliff — and on a forward that kept its
# state in the library, not in rast.bstate, a wrong one).  Materialise the cotangent once and say so.
function GaussianSplatting.∇rasterize(vpixels::AbstractArray{Float32, 3}, means_3d::RM, shs::R3, scales::RM, rotations::RM,
        opacities::RM, radii::ROCVector{Int32}, R_w2c = nothing, t_w2c = nothing;
        rast::GaussianRasterizer, camera::Camera, sh_degree::Int, background::SVector{3, Float32})
    native(rast) === nothing && return invoke(∇rasterize, REF_BWD_SIG, vpixels, means_3d, shs, scales, rotations, opacities,
        radii, R_w2c, t_w2c; rast, camera, sh_degree, background)
    @warn "GaussianSplattingHipNative: cotangent of type $(typeof(vpixels)) copied into a ROCArray for the native ∇rasterize" maxlog=1
    vp = similar(rast.image)
    vp .= vpixels
    return ∇rasterize(vp, means_3d, shs, scales, rotations, opacities, radii, R_w2c, t_w2c; rast, camera, sh_degree, background)
end

# rrule(rasterize, ...) — the reference's own (rasterizer.jl:552-573) with ROCArray positional types: the only addition is the
# note to the forward that this render WILL be differentiated (a bare `rasterize` is rendered forward-only).
function ChainRulesCore.rrule(::typeof(rasterize), means_3d::RM, shs::R3, opacities::RM, scales::RM, rotations::RM,
        R_w2c = nothing, t_w2c = nothing;
        rast::GaussianRasterizer, camera::Camera, sh_degree::Int, background::SVector{3, Float32},
        covisibilities = nothing, uncertainties = nothing)
    st = native(rast)
    st === nothing || (st.pullback_follows = true)
    image = rasterize(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c;
        rast, camera, sh_degree, background, covisibilities, uncertainties)
    function _pullback(vpixels)
        ∇ = ∇rasterize(unthunk(vpixels), means_3d, shs, scales, rotations, opacities,
            rast.gstate.radii, R_w2c, t_w2c; rast, camera, sh_degree, background)
        return (NoTangent(), ∇...)
    end
    return image, _pullback
end

# gsr_reserve: pre-size the native scratch for up to n_gaussians / n_instances (call it after a densification round, with headroom:
# a reallocation inside a forward synchronises the device in the middle of a training step)
function reserve!(rast::GaussianRasterizer, n_gaussians::Integer, n_instances::Integer = 0)
    st = native(rast)
    st === nothing || check(ccall((:gsr_reserve, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), st.handle, n_gaussians, n_instances))
    return rast
end

# release_scene_buffers!(rast) (rasterizer.jl:111-123) also has to drop the library's scene-sized scratch
function release_native_scene_buffers!(rast::GaussianRasterizer)
    st = native(rast)
    st === nothing || check(ccall((:gsr_release_scene_buffers, LIB), Cint, (Ptr{Cvoid},), st.handle))
    return
end

# ---- process-wide switches ----

# Form of the binning inside the forward's first kernel: -1 by scene and grid size (default), 0 direct, 1 aggregating wherever
# its LDS fits.  Identical outputs in every form (include/gsr.h: gsr_preprocess_form); a performance switch.
preprocess_form!(form::Integer) = check(ccall((:gsr_preprocess_form, LIB), Cint, (Cint,), form))
# Arithmetic of the SSIM entry points: false = contracted (default), true = every fp32 operation as written (gsr_ssim_precision)
ssim_exact!(exact::Bool) = check(ccall((:gsr_ssim_precision, LIB), Cint, (Cint,), exact ? 1 : 0))

# Optional (not a reference function): 63-bit Morton codes of the positions (3 x N Float32 device array) inside the box
# lo .. hi — sortperm of them is the permutation of a spatial re-sort (apply it to every per-Gaussian array, `gs.ids` included).
function morton_codes!(codes, points, lo::NTuple{3,Float32}, hi::NTuple{3,Float32})
    n = size(points, 2)
    check(ccall((:gsr_morton_codes, LIB), Cint, (Int64, Ptr{Float32}, Ref{NTuple{3,Float32}}, Ref{NTuple{3,Float32}}, Ptr{UInt64}, Ptr{Cvoid}),
                n, dptr(points), Ref(lo), Ref(hi), dptr(UInt64, codes), hipstream()))
    return codes
end

# ---- optional fused tails (not needed for the drop-in; callers opt in) ----

# update_stats!(strategy, radii, ∇means_2d, resolution) (strategy.jl:107-136) on the library's kernel
update_stats!(strategy, rast::GaussianRasterizer) = check(ccall((:gsr_update_stats, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}), native(rast).handle,
    dptr(Int32, strategy.max_radii), dptr(strategy.accum_∇means_2d), dptr(strategy.denom), hipstream()))

# The tail of `step!` (training.jl:768-779 + the prologue of the next `rast(...)` call, rasterizer.jl:218-247) in one
# pass: `∇` are the cotangents ∇rasterize returned (w.r.t. the ACTIVATED opacity / scale), `θ` the raw parameter
# arrays and `opts` the six NU.Adam in OPTIMIZER_NAMES order; shs / opacities_act / scales_act are the activated
# copies of this step on entry and of the updated parameters on exit.
struct GsrTailGrads; vmeans::Ptr{Float32}; vshs::Ptr{Float32}; vopacities::Ptr{Float32}; vscales::Ptr{Float32}; vrotations::Ptr{Float32}; end
function trainer_tail_step!(θ::NTuple{6}, opts::NTuple{6}, ∇, shs, opacities_act, scales_act; β1=0.9f0, β2=0.999f0, ϵ=1f-15)
    vmeans, vshs, vopac, vscales, vrot = ∇
    p(xs) = Ptr{Float32}[dptr(x) for x in xs]
    steps = UInt32[o.current_step + 0x1 for o in opts]  # the counters AFTER their increment, committed below
    check(ccall((:gsr_trainer_tail_step, LIB), Cint,
        (Cint, Cint, Cint, Ref{GsrTailGrads}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Cfloat}, Ptr{UInt32},
         Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
        size(θ[1], 2), size(θ[3], 2), size(θ[5], 1), GsrTailGrads(dptr(vmeans), dptr(vshs), dptr(vopac), dptr(vscales), dptr(vrot)),
        p(θ), p(map(o -> o.μ[1], opts)), p(map(o -> o.ν[1], opts)), Float32[o.lr for o in opts],
        steps, β1, β2, ϵ, dptr(shs), dptr(opacities_act), dptr(scales_act), hipstream()))
    foreach(o -> o.current_step += 0x1, opts)
end

# The same tail applied in the epilogue of the backward (gsr_backward_trainer_tail): the single-GPU `step!` after
# the loss with no gradient arrays at all.  `vpixels` is the loss cotangent of rast.image; θ[1] / θ[6] must be the
# arrays the forward was given as means_3d / rotations, shs / opacities_act / scales_act its other inputs.
# `color_cotangent = true` (per call, GSR_GRADS_COLOR_COTANGENT): `vpixels` is the buffer loss_l1_ssim! returned for this forward,
# untouched — in :rgbd / :rgbdn mode the backward then runs the :rgb arithmetic (:rgbdn 0.885 -> 0.702 ms at config 3); any other
# buffer with the flag set is an error, not silently dropped depth / normal gradients.
function backward_trainer_tail!(rast::GaussianRasterizer, vpixels, θ::NTuple{6}, opts::NTuple{6}, shs, opacities_act,
        scales_act; camera::Camera, sh_degree::Int, background::SVector{3, Float32}, β1=0.9f0, β2=0.999f0, ϵ=1f-15,
        color_cotangent::Bool = false)
    st = native(rast)
    st === nothing && error("backward_trainer_tail! needs enable_hip_native!(rast)")
    inp, cam = _structs(θ[1], shs, opacities_act, scales_act, θ[6], nothing, nothing, camera, sh_degree, background)
    p(xs) = ntuple(i -> dptr(xs[i]), 6)
    ts = GsrTailState(p(θ), p(map(o -> o.μ[1], opts)), p(map(o -> o.ν[1], opts)), ntuple(i -> Float32(opts[i].lr), 6),
        ntuple(i -> UInt32(opts[i].current_step + 0x1), 6), β1, β2, ϵ, size(θ[5], 1),
        dptr(shs), dptr(opacities_act), dptr(scales_act), Ptr{Float32}(UInt(pointer(rast.gstate.∇means_2d))), st.generation,
        color_cotangent ? GSR_GRADS_COLOR_COTANGENT : UInt32(0), UInt32(0))
    check(ccall((:gsr_backward_trainer_tail, LIB), Cint,
        (Ptr{Cvoid}, Ref{GsrInputs}, Ref{GsrCamera}, Ptr{Float32}, Ref{GsrTailState}, Ptr{Cvoid}),
        st.handle, inp, cam, dptr(vpixels), ts, hipstream()))
    foreach(o -> o.current_step += 0x1, opts)
end

end # module
