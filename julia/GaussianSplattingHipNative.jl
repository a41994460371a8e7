# Julia-side binding of libgsr_hip.so for GaussianSplatting.jl (reference @ v2.0.0).
#
# This file could not be executed in the build environment (no Julia there); it is kept
# small and mirrors, field for field, the ctypes binding that IS tested
# (gaussiansplatting.jl_amd/_lib.py + rasterizer.py).  It adds `rasterize` / `rrule`
# methods for a `HipNativeRasterizer`, so `Trainer.step!`, `validate`, the GUI worker etc.
# (callers listed in SURVEY.md §8b) run unchanged once they construct this rasterizer
# instead of `GaussianRasterizer`.
#
# Replaces: src/rasterization/rasterizer.jl:255-408 (rasterize), :416-550 (∇rasterize),
#           :552-573 (rrule).
module GaussianSplattingHipNative

using AMDGPU, ChainRulesCore, StaticArrays
import GaussianSplatting
import GaussianSplatting: Camera, resolution

const LIB = get(ENV, "GSR_HIP_LIB", "libgsr_hip.so")

struct GsrConfig
    width::Int32; height::Int32; mode::Int32
    near_plane::Float32; far_plane::Float32; radius_clip::Int32; blur_eps::Float32; flags::UInt32
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
struct GsrAux; covisibilities::Ptr{UInt8}; uncertainties::Ptr{Float32}; end
struct GsrGrads
    vmeans::Ptr{Float32}; vshs::Ptr{Float32}; vopacities::Ptr{Float32}
    vscales::Ptr{Float32}; vrotations::Ptr{Float32}; vR::Ptr{Float32}; vt::Ptr{Float32}
    vcolors::Ptr{Float32}  # C_NULL: classic form (∇shs written); see gsr.h for the factored multi-view form
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:gsr_last_error_string, LIB), Cstring, ())))
dptr(x) = x === nothing ? Ptr{Float32}(C_NULL) : Ptr{Float32}(UInt(pointer(x)))
hipstream() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))  # the task-local stream (gui/worker.jl:47-51)

mutable struct HipNativeRasterizer
    handle::Ptr{Cvoid}
    image::ROCArray{Float32, 3}
    mode::Symbol
    width::Int; height::Int
end

function HipNativeRasterizer(; width::Int, height::Int, mode::Symbol = :rgbd,
                             near_plane::Float32 = 0.2f0, far_plane::Float32 = 1000f0,
                             exact_tile_cull::Bool = false)
    c = GaussianSplatting.n_color_features(mode)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:gsr_create, LIB), Cint, (Ref{GsrConfig}, Ref{Ptr{Cvoid}}),
        GsrConfig(width, height, c, near_plane, far_plane, 3, 0.3f0, exact_tile_cull ? 1 : 0), h))
    r = HipNativeRasterizer(h[], AMDGPU.zeros(Float32, c, width, height), mode, width, height)
    finalizer(x -> ccall((:gsr_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), r)
    return r
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

function GaussianSplatting.rasterize(means_3d::ROCArray, shs::ROCArray, opacities::ROCArray, scales::ROCArray,
        rotations::ROCArray, R_w2c = nothing, t_w2c = nothing;
        rast::HipNativeRasterizer, camera::Camera, sh_degree::Int, background::SVector{3, Float32},
        covisibilities = nothing, uncertainties = nothing)
    inp, cam = _structs(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, camera, sh_degree, background)
    aux = GsrAux(covisibilities === nothing ? C_NULL : Ptr{UInt8}(UInt(pointer(covisibilities))), dptr(uncertainties))
    check(ccall((:gsr_forward, LIB), Cint,
        (Ptr{Cvoid}, Ref{GsrInputs}, Ref{GsrCamera}, Ptr{Float32}, Ref{GsrAux}, Ptr{Cvoid}, Ptr{Cvoid}),
        rast.handle, inp, cam, dptr(rast.image), aux, hipstream(), C_NULL))
    return rast.image
end

function ChainRulesCore.rrule(::typeof(GaussianSplatting.rasterize), means_3d::ROCArray, shs, opacities, scales,
        rotations, R_w2c = nothing, t_w2c = nothing; rast::HipNativeRasterizer, camera::Camera, sh_degree::Int,
        background::SVector{3, Float32}, covisibilities = nothing, uncertainties = nothing)
    image = GaussianSplatting.rasterize(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c;
        rast, camera, sh_degree, background, covisibilities, uncertainties)
    function _pullback(vpixels)
        vp = unthunk(vpixels)
        n = size(means_3d, 2)
        vmeans = similar(means_3d); vshs = similar(shs); vopac = similar(opacities)
        vscales = similar(scales); vrot = similar(rotations)
        vR = R_w2c === nothing ? nothing : AMDGPU.zeros(Float32, 3, 3)
        vt = R_w2c === nothing ? nothing : AMDGPU.zeros(Float32, 3)
        inp, cam = _structs(means_3d, shs, opacities, scales, rotations, R_w2c, t_w2c, camera, sh_degree, background)
        check(ccall((:gsr_backward, LIB), Cint,
            (Ptr{Cvoid}, Ref{GsrInputs}, Ref{GsrCamera}, Ptr{Float32}, Ref{GsrGrads}, Ptr{Cvoid}),
            rast.handle, inp, cam, dptr(vp),
            GsrGrads(dptr(vmeans), dptr(vshs), dptr(vopac), dptr(vscales), dptr(vrot), dptr(vR), dptr(vt), dptr(nothing)), hipstream()))
        return (NoTangent(), vmeans, vshs, vopac, vscales, vrot, vR, vt)
    end
    return image, _pullback
end

# Side outputs densification reads (src/strategy.jl:85-86): rast.gstate.radii / ∇means_2d
function state_buffer(rast::HipNativeRasterizer, which::Integer, ::Type{T}, dims) where T
    p = Ref{Ptr{Cvoid}}(); sz = Ref{Csize_t}()
    check(ccall((:gsr_buffer, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}, Ref{Csize_t}), rast.handle, which, p, sz))
    return unsafe_wrap(ROCArray, Ptr{T}(p[]), dims; own=false)
end
radii(rast, n) = state_buffer(rast, 0, Int32, (n,))
grad_means_2d(rast, n) = state_buffer(rast, 1, Float32, (2, n))

# update_stats!(strategy, radii, ∇means_2d, resolution) (src/strategy.jl:107-116) without the copies
update_stats!(strategy, rast::HipNativeRasterizer) = check(ccall((:gsr_update_stats, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}), rast.handle,
    Ptr{Int32}(UInt(pointer(strategy.max_radii))), dptr(strategy.accum_∇means_2d), dptr(strategy.denom), hipstream()))

# The tail of `step!` (src/training.jl:768-779 + the prologue of the next `rast(...)` call, rasterizer.jl:218-247)
# in one pass: `∇` are the cotangents the rrule above returned (w.r.t. the ACTIVATED opacity / scale), `θ` the raw
# parameter arrays and `opts` the six NU.Adam in OPTIMIZER_NAMES order; shs / opacities_act / scales_act are the
# activated copies of this step on entry and of the updated parameters on exit.
struct GsrTailGrads; vmeans::Ptr{Float32}; vshs::Ptr{Float32}; vopacities::Ptr{Float32}; vscales::Ptr{Float32}; vrotations::Ptr{Float32}; end
function trainer_tail_step!(θ::NTuple{6}, opts::NTuple{6}, ∇, shs, opacities_act, scales_act; β1=0.9f0, β2=0.999f0, ϵ=1f-15)
    vmeans, vshs, vopac, vscales, vrot = ∇
    foreach(o -> o.current_step += 0x1, opts)
    p(xs) = Ptr{Float32}[dptr(x) for x in xs]
    check(ccall((:gsr_trainer_tail_step, LIB), Cint,
        (Cint, Cint, Cint, Ref{GsrTailGrads}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Ptr{Float32}}, Ptr{Cfloat}, Ptr{UInt32},
         Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
        size(θ[1], 2), size(θ[3], 2), size(θ[5], 1), GsrTailGrads(dptr(vmeans), dptr(vshs), dptr(vopac), dptr(vscales), dptr(vrot)),
        p(θ), p(map(o -> o.μ[1], opts)), p(map(o -> o.ν[1], opts)), Float32[o.lr for o in opts],
        UInt32[o.current_step for o in opts], β1, β2, ϵ, dptr(shs), dptr(opacities_act), dptr(scales_act), hipstream()))
end

end # module
