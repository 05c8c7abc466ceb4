"""Scene recipes of the harness (numpy, float32): the reference's Cornell box, its box / sphere
meshes and flat materials, the synthetic N-triangle scenes of BASELINE.json's configs, the BRDF
LUT, cameras and lightmap pages.  Everything here produces INPUTS (a PimSceneDesc of host arrays);
the same desc is handed to libpimgpu and, in tests, to the CPU checker.

Reference recipes followed (paths relative to the Pim tree, src/):
  GenBoxMesh / GenSphereMesh       rendering/render_system.c:745-1018
  GenFlatTexture / GenMaterial     rendering/render_system.c:1020-1051
  CmdCornellBox                    rendering/render_system.c:1110-1305
  FlattenDrawables (TRS -> world)  rendering/path_tracer.c:692-782, math/float4x4_funcs.h:128-192
  pt_test camera                   rendering/render_system.c:1348-1372 (teleport -4 0 4; lookat 0 -1 0)
  media_desc_new / DofInfo_New     rendering/path_tracer.c:1944-1961, 1141-1153
"""
import ctypes
import math

import numpy as np

from . import api

F = np.float32
kPi = F(3.1415926535897932384626433832795)
kTau = F(6.283185307179586476925286766559)
kEpsSq = F(1.0) / F(70368744177664.0)
kDeci = F(0.1)

# Cubemap_kForwards / kUps, rendering/cubemap.c:14-32 (XP, XM, YP, YM, ZP, ZM)
K_FORWARDS = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], F)
K_UPS = np.array([[0, 1, 0], [0, 1, 0], [0, 0, -1], [0, 0, -1], [0, 1, 0], [0, 1, 0]], F)
XP, XM, YP, YM, ZP, ZM = range(6)


# ---------------------------------------------------------------------------------------------
# float32 vector helpers in the reference's operation order

def _v3(x, y, z):
    return np.array([x, y, z], F)


def dot3(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def cross3(a, b):
    # f4_cross3: zxy(zxy(a)*b - a*zxy(b)), math/float4_funcs.h:556
    azxy = _v3(a[2], a[0], a[1]); bzxy = _v3(b[2], b[0], b[1])
    t = (azxy * b - a * bzxy).astype(F)
    return _v3(t[2], t[0], t[1])


def normalize3(a):
    l = F(np.sqrt(np.maximum(kEpsSq, dot3(a, a))))
    return (a * (F(1.0) / l)).astype(F)


def quat_lookat(forward, up):
    """math/quat_funcs.h:223-233 + f3x3_quat :160-221; returns xyzw"""
    forward = (-np.asarray(forward, F)).astype(F)
    up = np.asarray(up, F)
    right = normalize3(cross3(up, forward))
    up = normalize3(cross3(forward, right))
    c0, c1, c2 = right, up, forward
    m00, m01, m02 = c0
    m10, m11, m12 = c1
    m20, m21, m22 = c2
    x = F(F(m00 - m11) - m22); y = F(F(m11 - m00) - m22); z = F(F(m22 - m00) - m11); w = F(F(m00 + m11) + m22)
    b = max(max(x, y), max(z, w))
    i = 0
    if b == x: i = 1
    if b == y: i = 2
    if b == z: i = 3
    b = F(F(np.sqrt(max(F(b + F(1.0)), F(4.0) * kEpsSq))) * F(0.5))
    mult = F(F(0.25) / b)
    if i == 0:
        r = [F(m12 - m21) * mult, F(m20 - m02) * mult, F(m01 - m10) * mult, b]
    elif i == 1:
        r = [b, F(m01 + m10) * mult, F(m20 + m02) * mult, F(m12 - m21) * mult]
    elif i == 2:
        r = [F(m01 + m10) * mult, b, F(m12 + m21) * mult, F(m20 - m02) * mult]
    else:
        r = [F(m20 + m02) * mult, F(m12 + m21) * mult, b, F(m01 - m10) * mult]
    return np.array(r, F)


QUAT_ID = np.array([0, 0, 0, 1], F)


def quat_to_m33(q):
    """quat_f3x3, math/quat_funcs.h:131-158; returns columns c0, c1, c2"""
    x, y, z, w = [F(v) for v in q]
    xx, xy, xz = F(x * x), F(x * y), F(x * z)
    yy, yz, zz = F(y * y), F(y * z), F(z * z)
    wx, wy, wz = F(w * x), F(w * y), F(w * z)
    one, two = F(1.0), F(2.0)
    c0 = _v3(one - two * F(yy + zz), two * F(xy - wz), two * F(xz + wy))
    c1 = _v3(two * F(xy + wz), one - two * F(xx + zz), two * F(yz - wx))
    c2 = _v3(two * F(xz - wy), two * F(yz + wx), one - two * F(xx + yy))
    return c0, c1, c2


def trs_matrix(t, q, s):
    """f4x4_trs, math/float4x4_funcs.h:128-192: columns c0..c3 (xyz)"""
    c0, c1, c2 = quat_to_m33(q)
    s = np.asarray(s, F)
    return (c0 * s[0]).astype(F), (c1 * s[1]).astype(F), (c2 * s[2]).astype(F), np.asarray(t, F)[:3].copy()


def flatten_entity(t, q, s, mesh):
    """FlattenDrawables for one entity, rendering/path_tracer.c:741-771: world positions (w=1),
    normals = normalize(IM * n) with IM = inverse(transpose(M3x3)), uvs"""
    pos, nor, uv = mesh
    c0, c1, c2, c3 = trs_matrix(t, q, s)
    # f4x4_mul_pt: (c0*x + c1*y) + (c2*z + c3)
    P = ((c0[None, :] * pos[:, 0:1] + c1[None, :] * pos[:, 1:2]).astype(F) + (c2[None, :] * pos[:, 2:3] + c3[None, :]).astype(F)).astype(F)
    # IM = inverse(transpose(M)); math/float3x3_funcs.h:71-110 ([column][row] accessors)
    tc0 = _v3(c0[0], c1[0], c2[0]); tc1 = _v3(c0[1], c1[1], c2[1]); tc2 = _v3(c0[2], c1[2], c2[2])
    m00, m01, m02 = tc0; m10, m11, m12 = tc1; m20, m21, m22 = tc2
    a = F(F(m00 * m11) - F(m10 * m01)); b = F(F(m00 * m12) - F(m10 * m02)); c = F(F(m00 * m21) - F(m20 * m01))
    d = F(F(m00 * m22) - F(m20 * m02)); e = F(F(m01 * m12) - F(m11 * m02)); f = F(F(m01 * m22) - F(m21 * m02))
    g = F(F(m10 * m21) - F(m20 * m11)); h = F(F(m10 * m22) - F(m20 * m12)); i = F(F(m11 * m22) - F(m21 * m12))
    det = F(F(F(m00 * i) + F(-m10 * f)) + F(m20 * e))
    r = F(F(1.0) / det)
    i0 = _v3(r * i, -r * f, r * e); i1 = _v3(-r * h, r * d, -r * b); i2 = _v3(r * g, -r * c, r * a)
    # f3x3_mul_col: c0*x + (c1*y + c2*z)
    N = (i0[None, :] * nor[:, 0:1] + (i1[None, :] * nor[:, 1:2] + i2[None, :] * nor[:, 2:3]).astype(F)).astype(F)
    l2 = ((N[:, 0] * N[:, 0] + N[:, 1] * N[:, 1]).astype(F) + N[:, 2] * N[:, 2]).astype(F)
    l = np.sqrt(np.maximum(kEpsSq, l2)).astype(F)
    N = (N * (F(1.0) / l)[:, None]).astype(F)
    P4 = np.concatenate([P, np.ones((P.shape[0], 1), F)], axis=1)
    N4 = np.concatenate([N, np.zeros((N.shape[0], 1), F)], axis=1)
    return P4, N4, uv.astype(F).copy()


# ---------------------------------------------------------------------------------------------
# meshes

def box_mesh():
    """GenBoxMesh, render_system.c:925-1018: 36 vertices, unit cube +-0.5"""
    v = np.array([[1, 1, -1], [1, -1, -1], [1, 1, 1], [1, -1, 1], [-1, 1, -1], [-1, -1, -1], [-1, 1, 1], [-1, -1, 1]], F)
    vt = np.array([[0.875, 0.5], [0.625, 0.75], [0.625, 0.5], [0.375, 1.0], [0.375, 0.75], [0.625, 0.0], [0.375, 0.25],
                   [0.375, 0.0], [0.375, 0.5], [0.125, 0.75], [0.125, 0.5], [0.625, 0.25], [0.875, 0.75], [0.625, 1.0]], F)
    vn = np.array([[0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, -1]], F)
    f = np.array([[5, 1, 1], [3, 2, 1], [1, 3, 1], [3, 2, 2], [8, 4, 2], [4, 5, 2], [7, 6, 3], [6, 7, 3], [8, 8, 3],
                  [2, 9, 4], [8, 10, 4], [6, 11, 4], [1, 3, 5], [4, 5, 5], [2, 9, 5], [5, 12, 6], [2, 9, 6], [6, 7, 6],
                  [5, 1, 1], [7, 13, 1], [3, 2, 1], [3, 2, 2], [7, 14, 2], [8, 4, 2], [7, 6, 3], [5, 12, 3], [6, 7, 3],
                  [2, 9, 4], [4, 5, 4], [8, 10, 4], [1, 3, 5], [3, 2, 5], [4, 5, 5], [5, 12, 6], [1, 3, 6], [2, 9, 6]]) - 1
    return (v[f[:, 0]] * F(0.5)).astype(F), vn[f[:, 2]].copy(), vt[f[:, 1]].copy()


def sphere_mesh(vsteps=24):
    """GenSphereMesh, render_system.c:745-876: vsteps x 2*vsteps UV sphere, radius 1, normals = positions"""
    hsteps = vsteps * 2
    dv = F(kPi / F(vsteps)); dh = F(kTau / F(hsteps))
    P, U = [], []
    for v in range(vsteps):
        th1 = F(F(v) * dv); th2 = F(F(v + 1) * dv)
        st1, ct1, st2, ct2 = F(math.sin(th1)), F(math.cos(th1)), F(math.sin(th2)), F(math.cos(th2))
        for h in range(hsteps):
            ph1 = F(F(h) * dh); ph2 = F(F(h + 1) * dh)
            sp1, cp1, sp2, cp2 = F(math.sin(ph1)), F(math.cos(ph1)), F(math.sin(ph2)), F(math.cos(ph2))
            u1 = (F(ph1 / kTau), F(F(1.0) - F(th1 / kPi))); u2 = (F(ph2 / kTau), F(F(1.0) - F(th1 / kPi)))
            u3 = (F(ph2 / kTau), F(F(1.0) - F(th2 / kPi))); u4 = (F(ph1 / kTau), F(F(1.0) - F(th2 / kPi)))
            n1 = (F(st1 * cp1), ct1, F(st1 * sp1)); n2 = (F(st1 * cp2), ct1, F(st1 * sp2))
            n3 = (F(st2 * cp2), ct2, F(st2 * sp2)); n4 = (F(st2 * cp1), ct2, F(st2 * sp1))
            if v == 0:
                P += [n1, n3, n4]; U += [u1, u3, u4]
            elif v + 1 == vsteps:
                P += [n3, n1, n2]; U += [u3, u1, u2]
            else:
                P += [n1, n2, n4, n2, n3, n4]; U += [u1, u2, u4, u2, u3, u4]
    P = np.array(P, F)
    return P, P.copy(), np.array(U, F)


# ---------------------------------------------------------------------------------------------
# textures and materials

def gamma_encode_rgba8(v):
    """GammaEncode_rgba8 = f4_rgba8(f4_sRGB_InverseEOTF_Fit(v)), math/color.h:102-112,212-218,251"""
    v = np.asarray(v, F)
    l1 = np.sqrt(v).astype(F); l2 = np.sqrt(l1).astype(F); l3 = np.sqrt(l2).astype(F)
    e = ((F(0.658444) * l1 + F(0.643378) * l2).astype(F) + F(-0.298148) * l3).astype(F)
    e = (np.clip(e, F(0), F(1)) * F(255.0) + F(0.5)).astype(F)
    b = e.astype(np.uint32) & 0xff
    return np.uint32(int(b[0]) | (int(b[1]) << 8) | (int(b[2]) << 16) | (int(b[3]) << 24))


class SceneBuilder:
    """accumulates entities (mesh + TRS + material) and flattens them into a PimSceneDesc"""

    def __init__(self, seed=1, sky_size=0, emissive_attempts=1000, meters_per_cell=1.5):
        self.pos, self.nor, self.uv, self.mat = [], [], [], []
        self.materials, self.textures = [], []
        self.seed, self.sky_size = seed, sky_size
        self.emissive_attempts, self.meters_per_cell = emissive_attempts, meters_per_cell
        self.media = default_media()
        self.lut = None
        self._flat = {}
        self.entities = []          # (mesh, t, q, s, material index) of every add(): the draw table before flattening

    def flat_texture(self, value):
        key = tuple(np.asarray(value, F).tolist())
        if key not in self._flat:
            self.textures.append((1, 1, api.TEX_RGBA8, np.array([gamma_encode_rgba8(value)], np.uint32)))
            self._flat[key] = len(self.textures) - 1
        return self._flat[key]

    def add_texture(self, width, height, fmt, texels):
        self.textures.append((width, height, fmt, np.ascontiguousarray(texels, np.uint32).reshape(-1)))
        return len(self.textures) - 1

    def material(self, albedo, rome, flags=0, ior=1.0, albedo_tex=None, rome_tex=None, normal_tex=-1):
        """GenMaterial, render_system.c:1034-1051 (Color_SDRToScene is the identity: COLOR_SCENE_REC709)"""
        m = api.PimMaterial()
        m.albedo = self.flat_texture(albedo) if albedo_tex is None else albedo_tex
        m.rome = self.flat_texture(rome) if rome_tex is None else rome_tex
        m.normal = normal_tex
        m.flags = flags | (api.MATFLAG_EMISSIVE if rome[3] > 0.0 else 0)
        m.ior = ior
        self.materials.append(m)
        return len(self.materials) - 1

    def add(self, mesh, t, q, s, mat, uv_scale=None, uv_offset=None):
        if uv_scale is not None or uv_offset is not None:
            # a mesh of its own with uvs' = uvs * scale + offset (float32): texture coordinates outside [0, 1] and negative
            sc_ = np.asarray(uv_scale if uv_scale is not None else (1.0, 1.0), F); of_ = np.asarray(uv_offset if uv_offset is not None else (0.0, 0.0), F)
            mesh = (mesh[0].copy(), mesh[1].copy(), ((mesh[2] * sc_[None, :]).astype(F) + of_[None, :]).astype(F))
        P, N, U = flatten_entity(t, q, s, mesh)
        self.entities.append((mesh, np.asarray(t, F), np.asarray(q, F), np.asarray(s, F), mat))
        self.pos.append(P); self.nor.append(N); self.uv.append(U)
        self.mat.append(np.full(P.shape[0], mat, np.int32))

    def add_arrays(self, P4, N4, U, mat):
        self.pos.append(np.ascontiguousarray(P4, F)); self.nor.append(np.ascontiguousarray(N4, F)); self.uv.append(np.ascontiguousarray(U, F))
        self.mat.append(np.full(P4.shape[0], mat, np.int32) if np.isscalar(mat) else np.ascontiguousarray(mat, np.int32))

    def build(self):
        return SceneData(self)


class SceneData:
    """owns the numpy arrays a PimSceneDesc points into"""

    def __init__(self, b):
        self.positions = np.ascontiguousarray(np.concatenate(b.pos), F) if b.pos else np.zeros((0, 4), F)
        self.normals = np.ascontiguousarray(np.concatenate(b.nor), F) if b.nor else np.zeros((0, 4), F)
        self.uvs = np.ascontiguousarray(np.concatenate(b.uv), F) if b.uv else np.zeros((0, 2), F)
        self.matIds = np.ascontiguousarray(np.concatenate(b.mat), np.int32) if b.mat else np.zeros(0, np.int32)
        self.materials = (api.PimMaterial * max(1, len(b.materials)))(*b.materials)
        self.tex_arrays = [t[3] for t in b.textures]
        self.textures = (api.PimTexture * max(1, len(b.textures)))()
        for i, (w, h, fmt, texels) in enumerate(b.textures):
            self.textures[i].width, self.textures[i].height, self.textures[i].format = w, h, fmt
            self.textures[i].texels = texels.ctypes.data
        self.lut = b.lut if b.lut is not None else brdf_lut()
        self.entities = b.entities
        self.builder_materials = b.materials
        self.sky = sky_faces(b.sky_size) if b.sky_size > 0 else None
        d = api.PimSceneDesc()
        d.vertCount = self.positions.shape[0]
        d.matCount = len(b.materials)
        d.texCount = len(b.textures)
        d.skySize = b.sky_size
        d.positions = self.positions.ctypes.data
        d.normals = self.normals.ctypes.data
        d.uvs = self.uvs.ctypes.data
        d.matIds = self.matIds.ctypes.data
        d.materials = ctypes.cast(self.materials, ctypes.c_void_p)
        d.textures = ctypes.cast(self.textures, ctypes.c_void_p)
        if self.sky is not None:
            for f in range(6):
                d.skyFaces[f] = self.sky[f].ctypes.data
        d.brdfLut = self.lut.ctypes.data
        d.brdfLutSize = self.lut.shape[0]
        d.emissiveAttempts = b.emissive_attempts
        d.metersPerCell = b.meters_per_cell
        d.seed = b.seed
        d.media = b.media
        self.desc = d

    @property
    def tri_count(self):
        return self.positions.shape[0] // 3


def drawables_of(sc):
    """the draw table (Entities columns + mesh table, rendering/drawable.h:14-27) of a scene made with SceneBuilder.add only:
    what FlattenDrawables reads.  Each entity carries its own copy of its material, as Entities.materials does."""
    meshes, index, ids, mats = [], {}, [], []
    M = np.zeros((len(sc.entities), 4, 4), F)
    for i, (mesh, t, q, s, mat) in enumerate(sc.entities):
        if id(mesh[0]) not in index:
            pos, nor, uv = mesh
            n = pos.shape[0]
            P = np.zeros((n, 4), F); P[:, :3] = pos; P[:, 3] = 1.0
            N = np.zeros((n, 4), F); N[:, :3] = nor
            U = np.zeros((n, 4), F); U[:, :2] = uv
            index[id(mesh[0])] = len(meshes)
            meshes.append((P, N, U))
        ids.append(index[id(mesh[0])])
        c = trs_matrix(t, q, s)
        for k in range(4):
            M[i, k, :3] = c[k]
        M[i, 3, 3] = 1.0
        mats.append(sc.builder_materials[mat])
    return api.Drawables(meshes, np.array(ids, np.int32), M, mats)


def default_media():
    """media_desc_new, rendering/path_tracer.c:1944-1961"""
    m = api.PimMediaDesc()
    m.constantColor[:] = [0.5, 0.5, 0.5, 2.0]
    m.noiseColor[:] = [0.5, 0.5, 0.5, 2.0]
    m.constantMfp = 40.0 * 1e3
    m.noiseMfp = 40.0 * 1e3
    m.absorption = 0.1
    m.noiseOctaves = 1
    m.noiseGain = 0.9
    m.noiseLacunarity = 2.0666
    m.noiseFreq = 1.0
    m.noiseHeight = 20.0
    m.noiseScale = 1.0
    m.phaseDirA = 0.0
    m.phaseDirB = 0.0
    m.phaseBlend = 0.5
    return m


def foggy_media(mfp=8.0):
    """a dense medium (for the tests that exercise ScatterRay / CalcTransmittance / SamplePhaseDir)"""
    m = default_media()
    m.constantMfp = mfp
    m.noiseMfp = mfp
    m.noiseHeight = 0.0
    m.noiseScale = 4.0
    m.noiseOctaves = 2
    m.phaseDirA = 0.5
    m.phaseDirB = -0.2
    return m


def default_dof():
    """DofInfo_New, rendering/path_tracer.c:1141-1153"""
    d = api.PimDofInfo()
    d.aperture = float(F(5.0) * F(1e-3))      # 5.0f * kMilli in fp32
    d.focalLength = 6.0
    d.bladeCount = 5
    d.bladeRot = float(kPi / F(10.0))
    d.focalPlaneCurvature = 0.05
    d.autoFocus = 1
    d.autoFocusSpeed = 3.0
    return d


def camera(position, lookat_dir, up=(0, 1, 0), fovy=90.0):
    """Camera (rendering/camera.h:8-15); `lookat x y z` console command = quat_lookat(normalize(dir), up)"""
    c = api.PimCamera()
    c.position[:] = [position[0], position[1], position[2], 0.0]
    q = quat_lookat(normalize3(np.asarray(lookat_dir, F)), np.asarray(up, F))
    c.rotation[:] = [float(v) for v in q]
    c.zNear, c.zFar, c.fovy = 0.1, 500.0, fovy
    return c


def pt_test_camera():
    """CmdPtTest, render_system.c:1348-1372: teleport -4 0 4; lookat 0 -1 0 (direction = target - position)"""
    pos = _v3(-4, 0, 4)
    target = _v3(0, -1, 0)
    return camera(pos, (target - pos).astype(F))


def trace_params(sample_index, seed=0, max_bounces=0, update_dists=1, frame_dt=1.0 / 60.0):
    p = api.PimTraceParams()
    p.sampleWeight = 1.0 / (sample_index + 1)
    p.frameDt = frame_dt
    p.sampleIndex = sample_index
    p.seed = seed
    p.maxBounces = max_bounces
    p.updateDists = update_dists
    return p


_LUT_CACHE = {}


def brdf_lut(size=32, samples=64, seed=7):
    """BakeBrdfFn, math/lighting.c:31-84, with a fixed numpy stream (the reference's LUT is baked from a
    clock-seeded stream and refined every frame: SURVEY A.4); float2[size*size] = (DFV.x, DFV.y)"""
    key = (size, samples, seed)
    if key in _LUT_CACHE:
        return _LUT_CACHE[key]
    rng = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    res = np.zeros((size, size, 2), np.float64)
    eps = 1.0 / (2 ** 23 - 1)
    for _ in range(samples):
        xi = rng.random((size, size, 5))
        ang = xi[..., 0] * 2 * np.pi
        rad = 0.5 * np.sqrt(-np.log(np.maximum(1.0 - xi[..., 1], eps)))
        NoV = np.clip((xs + np.cos(ang) * rad) / size, eps, 1 - eps)
        alpha = np.clip((ys + np.sin(ang) * rad) / size, 1.0 / 1024, 1.0)
        phi = xi[..., 2] * 2 * np.pi
        sinV = np.sqrt(np.maximum(1 - NoV * NoV, 0))
        V = np.stack([sinV * np.cos(phi), sinV * np.sin(phi), NoV], -1)
        a2 = alpha * alpha
        bb = np.maximum(1 + (a2 - 1) * xi[..., 4], eps)
        cosT = np.sqrt(np.maximum((1 - xi[..., 4]) / bb, 0))
        sinT = np.sqrt(np.maximum(1 - cosT * cosT, 0))
        ph = xi[..., 3] * 2 * np.pi
        H = np.stack([sinT * np.cos(ph), sinT * np.sin(ph), cosT], -1)
        HoV = (H * V).sum(-1)
        L = 2 * HoV[..., None] * H - V
        NoL, NoH = L[..., 2], H[..., 2]
        f = 1 + (a2 - 1) * NoH * NoH
        D = a2 / np.maximum(f * f * np.pi, eps)
        pdf = D * NoH / np.maximum(4 * HoV, eps)
        ok = (NoL > eps) & (pdf > eps)
        v = NoL * np.sqrt(np.maximum(a2 + (NoV - NoV * a2) * NoV, 0)); l = NoV * np.sqrt(np.maximum(a2 + (NoL - NoL * a2) * NoL, 0))
        G = 0.5 / np.maximum(v + l, eps)
        # F_Dielectric(HoV, 1.000293, 1.52)
        ci = np.clip(HoV, 0, 1); si = np.sqrt(np.maximum(1 - ci * ci, 0)); st = 1.000293 / 1.52 * si
        ct = np.sqrt(np.maximum(1 - st * st, 0))
        rp = (1.52 * ci - 1.000293 * ct) / (1.52 * ci + 1.000293 * ct); rs = (1.000293 * ci - 1.52 * ct) / (1.000293 * ci + 1.52 * ct)
        Fc = np.clip((rp * rp + rs * rs) * 0.5, 0, 1)
        dg = np.where(ok, D / np.maximum(pdf, eps) * G * NoL, 0.0) / samples
        res[..., 0] += dg * Fc
        res[..., 1] += dg
    lut = np.ascontiguousarray(res.reshape(size * size, 2), F).reshape(size, size, 2)
    _LUT_CACHE[key] = lut
    return lut


def sky_faces(size):
    """a smooth procedural sky (stand-in for BakeSky's atmosphere cubemap, render_system.c:427-494):
    float3[size*size] per face, brighter towards +Y"""
    faces = []
    for f in range(6):
        ys, xs = np.meshgrid((np.arange(size) + 0.5) / size * 2 - 1, (np.arange(size) + 0.5) / size * 2 - 1, indexing="ij")
        fwd = K_FORWARDS[f]; up = K_UPS[f]
        right = np.cross(up, fwd) if f not in (YP, YM) else np.array([1, 0, 0], F) * (1 if f == YP else -1)
        d = fwd[None, None, :] + xs[..., None] * right[None, None, :] + ys[..., None] * up[None, None, :]
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        h = np.clip(d[..., 1] * 0.5 + 0.5, 0, 1)
        col = np.stack([0.3 + 0.4 * h, 0.4 + 0.5 * h, 0.6 + 0.9 * h], -1) * 2.0
        faces.append(np.ascontiguousarray(col.reshape(-1, 3), F))
    return faces


# ---------------------------------------------------------------------------------------------
# scenes

def cornell_box(prims="boxes", seed=1, sky_size=0, media=None, emissive_attempts=1000):
    """CmdCornellBox, rendering/render_system.c:1110-1305"""
    b = SceneBuilder(seed=seed, sky_size=sky_size, emissive_attempts=emissive_attempts)
    if media is not None:
        b.media = media
    box, wallExtents = box_mesh(), F(5.0)
    wallScale = _v3(F(2.0) * wallExtents, F(2.0) * wallExtents, kDeci)
    cHi = F(0.9); cLo = F(F(1.0) - cHi)
    red, green, blue, white = (cHi, cLo, cLo, 1.0), (cLo, cHi, cLo, 1.0), (cLo, cLo, cHi, 1.0), (cHi, cHi, cHi, 1.0)
    plastic, metal, light = (0.9, 1.0, 0.0, 0.0), (0.1, 1.0, 1.0, 0.0), (0.9, 1.0, 0.0, 1.0)
    fw, up_ = K_FORWARDS, K_UPS

    def wall(face_t, face_look, color):
        b.add(box, (fw[face_t] * wallExtents).astype(F), quat_lookat(fw[face_look], up_[face_look]), wallScale, b.material(color, plastic))

    wall(YM, YP, white)                                     # Cornell_Floor
    wall(YP, YM, white)                                     # Cornell_Ceil
    b.add(box, (fw[YP] * F(wallExtents - kDeci * F(2.0))).astype(F), quat_lookat(fw[YM], up_[YM]),
          _v3(1.0, 1.0, kDeci), b.material((1.0, 1.0, 1.0, 1.0), light))    # Cornell_Light
    wall(XM, XP, green)                                     # Cornell_Left
    wall(XP, XM, red)                                       # Cornell_Right
    b.add(box, (fw[ZP] * wallExtents).astype(F), quat_lookat(fw[ZP], up_[ZP]), wallScale, b.material(white, plastic))   # Cornell_Near
    b.add(box, (fw[ZM] * wallExtents).astype(F), quat_lookat(fw[ZM], up_[ZM]), wallScale, b.material(blue, plastic))    # Cornell_Far

    def lerp(a, c, t):
        return F(F(a) + F(F(c) - F(a)) * F(t))

    if prims == "spheres":
        sph = sphere_mesh()
        sphereScale = F(0.75); margin = F(sphereScale * F(1.5))
        lo = F(-wallExtents + margin); hi = F(wallExtents - margin)
        flags = [0, 0, api.MATFLAG_REFRACTIVE]; metallics = [1.0, 0.0, 0.0]; iors = [1.0, 1.0, 1.5]
        for iRow in range(3):
            z = lerp(lo, hi, F(F(iRow + 0.5) / F(3)))
            for iCol in range(5):
                tCol = F(F(iCol + 0.5) / F(5))
                roughness = lerp(0.0, 1.0, tCol)
                x = lerp(lo, hi, tCol)
                m = b.material(white, (float(roughness), 1.0, metallics[iRow], 0.0), flags[iRow], ior=iors[iRow])
                b.add(sph, _v3(x, lo, z), QUAT_ID, _v3(sphereScale, sphereScale, sphereScale), m)
    else:
        boxScale = F(2.0); margin = F(boxScale * F(0.5))
        lo = F(-wallExtents + margin); hi = F(wallExtents - margin)
        b.add(box, _v3(lerp(lo, hi, 0.2), F(-wallExtents + boxScale), lerp(lo, hi, 0.2)),
              quat_lookat(normalize3(_v3(0.2, 0.0, 1.0)), _v3(0, 1, 0)), _v3(boxScale, F(boxScale * F(2.0)), boxScale),
              b.material(white, metal))                     # Cornell_MetalBox
        b.add(box, _v3(lerp(lo, hi, 0.8), F(-wallExtents + F(boxScale * F(0.5))), lerp(lo, hi, 0.8)),
              quat_lookat(normalize3(_v3(-0.2, 0.0, 1.0)), _v3(0, 1, 0)), _v3(boxScale, boxScale, boxScale),
              b.material(white, plastic))                   # Cornell_PlasticBox
    return b.build()


def pack_xy16(x, y):
    """short2 texels of a tangent-space normal map (Xy16ToNormalTs, math/color.h:241-249): x in the low half"""
    x = np.asarray(x, np.int32) & 0xffff; y = np.asarray(y, np.int32) & 0xffff
    return (x | (y << 16)).astype(np.uint32)


def textured_cornell(seed=1, emissive_attempts=1000, sky_size=0):
    """CmdCornellBox's room (render_system.c:1110-1305) with REAL textures instead of GenFlatTexture's 1x1 ones, to drive
    SampleAlbedo / SampleRome / SampleNormal (path_tracer.c:1328-1375), UvBilinearWrap_c32 / _xy16 (sampler.h:375-392)
    and EmissionPdf (path_tracer.c:784-825) through every branch:
      - non-square multi-texel rgba8 albedo (13x7) and ROME (10x10, 5x9) textures, xy16 normal maps (9x6, 4x11);
      - uvs scaled and shifted outside [0, 1] and below 0 (the wrap's `u < 0 -> 1 - u` branch);
      - a ROME whose alpha is non-zero on ONE texel of 100, on walls whose two triangles each cover half of the uv
        square several times over: their EmissionPdf lands around the 1 % threshold (path_tracer.c:869);
      - a ROME with an emissive stripe on a wall (part of each triangle emits), on top of the flat Cornell light."""
    rng = np.random.default_rng(seed + 77)
    b = SceneBuilder(seed=seed, sky_size=sky_size, emissive_attempts=emissive_attempts)
    box, wallExtents = box_mesh(), F(5.0)
    wallScale = _v3(F(2.0) * wallExtents, F(2.0) * wallExtents, kDeci)
    fw, up_ = K_FORWARDS, K_UPS

    def rgba8(r, g, b_, a):
        return (np.asarray(r, np.uint32) | (np.asarray(g, np.uint32) << 8) | (np.asarray(b_, np.uint32) << 16) | (np.asarray(a, np.uint32) << 24)).astype(np.uint32)

    # albedo 13 x 7: checker of two random palettes, alpha 255
    ys, xs = np.meshgrid(np.arange(7), np.arange(13), indexing="ij")
    chk = ((xs + ys) & 1).astype(bool)
    alb = rgba8(np.where(chk, rng.integers(150, 256, (7, 13)), rng.integers(20, 120, (7, 13))),
                rng.integers(60, 256, (7, 13)), np.where(chk, rng.integers(20, 90, (7, 13)), rng.integers(120, 256, (7, 13))), 255)
    albTex = b.add_texture(13, 7, api.TEX_RGBA8, alb)
    # ROME 10 x 10: roughness / occlusion / metallic vary, alpha non-zero on one texel of 100
    a1 = np.zeros((10, 10), np.uint32); a1[6, 3] = 200
    rome1 = rgba8(rng.integers(30, 256, (10, 10)), rng.integers(180, 256, (10, 10)), rng.integers(0, 2, (10, 10)) * 255, a1)
    romeSpot = b.add_texture(10, 10, api.TEX_RGBA8, rome1)
    # ROME 5 x 9: an emissive stripe (two texel rows of nine)
    a2 = np.zeros((9, 5), np.uint32); a2[3:5, :] = rng.integers(90, 256, (2, 5))
    rome2 = rgba8(rng.integers(100, 256, (9, 5)), 255, 0, a2)
    romeStripe = b.add_texture(5, 9, api.TEX_RGBA8, rome2)
    # ROME 3 x 2: smooth metal / rough plastic patches, no emission
    rome3 = rgba8(np.array([[25, 230, 120], [200, 40, 90]]), 255, np.array([[255, 0, 255], [0, 255, 0]]), 0)
    romeMetal = b.add_texture(3, 2, api.TEX_RGBA8, rome3)
    # normal maps
    n1 = pack_xy16(rng.integers(-9000, 9001, (6, 9)), rng.integers(-9000, 9001, (6, 9)))
    nrm1 = b.add_texture(9, 6, api.TEX_XY16, n1)
    n2 = pack_xy16(rng.integers(-20000, 20001, (11, 4)), rng.integers(-20000, 20001, (11, 4)))      # steep: FixShadingNormal's reflect branch
    nrm2 = b.add_texture(4, 11, api.TEX_XY16, n2)

    white, plastic, light = (0.9, 0.9, 0.9, 1.0), (0.9, 1.0, 0.0, 0.0), (0.9, 1.0, 0.0, 1.0)

    def mat(albedo_tex=None, rome_tex=None, normal_tex=-1, rome=plastic, flags=0):
        return b.material(white, rome, flags, albedo_tex=albedo_tex, rome_tex=rome_tex, normal_tex=normal_tex)

    def wall(face_t, face_look, m, sc=None, of=None):
        b.add(box, (fw[face_t] * wallExtents).astype(F), quat_lookat(fw[face_look], up_[face_look]), wallScale, m, uv_scale=sc, uv_offset=of)

    wall(YM, YP, mat(albTex, None, nrm1), (3.7, -2.3), (-1.3, 0.6))                      # floor: albedo + normal map, negative uvs
    wall(YP, YM, mat(None, romeSpot, -1, flags=api.MATFLAG_EMISSIVE), (2.0, 3.0), (0.25, -4.5))   # ceiling: ~1 % emissive texels
    b.add(box, (fw[YP] * F(wallExtents - kDeci * F(2.0))).astype(F), quat_lookat(fw[YM], up_[YM]),
          _v3(1.0, 1.0, kDeci), b.material((1.0, 1.0, 1.0, 1.0), light))                # Cornell_Light (flat)
    wall(XM, XP, mat(albTex, romeStripe, -1, flags=api.MATFLAG_EMISSIVE), (1.0, 1.0), (0.0, 0.0))   # left: emissive stripe
    wall(XP, XM, mat(albTex, romeMetal, nrm2), (-5.1, 1.9), (0.3, -0.2))                 # right: metal patches + steep normal map
    wall(ZP, ZP, mat(None, romeSpot, nrm1, flags=api.MATFLAG_EMISSIVE), (0.37, 0.41), (7.2, 100.6))  # near: spot ROME seen through few texels
    wall(ZM, ZM, mat(albTex, None, -1), (1.0, 1.0), (0.0, 0.0))                          # far: albedo only

    def lerp(a, c, t):
        return F(F(a) + F(F(c) - F(a)) * F(t))
    boxScale = F(2.0); margin = F(boxScale * F(0.5))
    lo = F(-wallExtents + margin); hi = F(wallExtents - margin)
    b.add(box, _v3(lerp(lo, hi, 0.2), F(-wallExtents + boxScale), lerp(lo, hi, 0.2)),
          quat_lookat(normalize3(_v3(0.2, 0.0, 1.0)), _v3(0, 1, 0)), _v3(boxScale, F(boxScale * F(2.0)), boxScale),
          mat(albTex, romeMetal, nrm1), uv_scale=(2.5, 2.5), uv_offset=(-0.75, -0.1))
    b.add(box, _v3(lerp(lo, hi, 0.8), F(-wallExtents + F(boxScale * F(0.5))), lerp(lo, hi, 0.8)),
          quat_lookat(normalize3(_v3(-0.2, 0.0, 1.0)), _v3(0, 1, 0)), _v3(boxScale, boxScale, boxScale),
          mat(albTex, romeStripe, nrm2, flags=api.MATFLAG_EMISSIVE), uv_scale=(1.0, -1.0), uv_offset=(0.0, 0.0))
    sph = sphere_mesh(12)
    b.add(sph, _v3(0.0, -3.5, 1.5), QUAT_ID, _v3(1.2, 1.2, 1.2), mat(albTex, romeMetal, nrm1), uv_scale=(4.0, 2.0), uv_offset=(-2.0, -1.0))
    return b.build()


def _random_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)


def _batch_flatten(mesh, T, Q, S):
    """vectorised TRS of many instances of one mesh (synthetic scenes only: plain float32 numpy)"""
    pos, nor, uv = mesh
    x, y, z, w = Q[:, 0], Q[:, 1], Q[:, 2], Q[:, 3]
    R = np.empty((Q.shape[0], 3, 3), F)   # R[:, :, k] = column k
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 1, 0] = 2 * (x * y - w * z); R[:, 2, 0] = 2 * (x * z + w * y)
    R[:, 0, 1] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 2, 1] = 2 * (y * z - w * x)
    R[:, 0, 2] = 2 * (x * z - w * y); R[:, 1, 2] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    M = R * S[:, None, :]
    P = np.einsum("nij,vj->nvi", M, pos).astype(F) + T[:, None, :]
    IM = np.linalg.inv(np.transpose(M, (0, 2, 1)).astype(np.float64)).astype(F)
    N = np.einsum("nij,vj->nvi", IM, nor).astype(F)
    N = N / np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
    n, v = P.shape[0], P.shape[1]
    P4 = np.concatenate([P.reshape(n * v, 3), np.ones((n * v, 1), F)], 1).astype(F)
    N4 = np.concatenate([N.reshape(n * v, 3), np.zeros((n * v, 1), F)], 1).astype(F)
    U = np.tile(uv, (n, 1)).astype(F)
    return P4, N4, U


SYNTH_CAMERA_POS = (-0.4, 0.05, 0.4)      # x extent: where synthetic_camera() stands; kept clear of objects
SYNTH_CAMERA_DIR = (1.0, -0.2, -1.0)


def synthetic_camera(extent=100.0):
    return camera(tuple(c * extent for c in SYNTH_CAMERA_POS), SYNTH_CAMERA_DIR)


def _keep_clear(T, extent):
    """mirror the objects that would swallow the camera position through the origin"""
    cam = np.array(SYNTH_CAMERA_POS, F) * F(extent)
    near = np.linalg.norm(T - cam[None, :], axis=1) < 0.04 * extent
    T[near] = -T[near]
    return T


def synthetic_scene(target_tris=1_000_000, seed=1234, extent=100.0, sphere_fraction=0.5, n_materials=64,
                    diffuse_only=False, emissive_attempts=64, meters_per_cell=None, sphere_vsteps=24):
    """BASELINE.json configs 3-5: random boxes (12 tris) and UV spheres (reference tessellation) in an
    `extent`-metre closed room with an emissive ceiling panel; flattened, no instancing.
    Deterministic in (target_tris, seed)."""
    rng = np.random.default_rng(seed)
    mpc = meters_per_cell if meters_per_cell is not None else max(1.5, extent / 48.0)
    b = SceneBuilder(seed=seed, emissive_attempts=emissive_attempts, meters_per_cell=mpc)
    box, sph = box_mesh(), sphere_mesh(sphere_vsteps)
    sph_tris = sph[0].shape[0] // 3
    half = F(extent * 0.5)
    # room: six wall boxes + light panel, as in the Cornell recipe but scaled
    wallScale = _v3(extent, extent, 0.1 * extent / 10.0)
    wallMat = b.material((0.7, 0.7, 0.7, 1.0), (1.0 if diffuse_only else 0.9, 1.0, 0.0, 0.0))
    wt = int(__import__("os").environ.get("PIM_WALL_TESS", "1"))      # experiment knob: tessellate the walls into wt x wt tiles
    for ft, fl in [(YM, YP), (YP, YM), (XM, XP), (XP, XM), (ZP, ZP), (ZM, ZM)]:
        if wt <= 1:
            b.add(box, (K_FORWARDS[ft] * half).astype(F), quat_lookat(K_FORWARDS[fl], K_UPS[fl]), wallScale, wallMat)
        else:
            q = quat_lookat(K_FORWARDS[fl], K_UPS[fl])
            c0, c1, c2 = quat_to_m33(q)
            for iy in range(wt):
                for ix in range(wt):
                    off = c0 * F((ix + 0.5) / wt - 0.5) * F(extent) + c1 * F((iy + 0.5) / wt - 0.5) * F(extent)
                    b.add(box, (K_FORWARDS[ft] * half + off).astype(F), q, _v3(extent / wt, extent / wt, wallScale[2]), wallMat)
    lightMat = b.material((1.0, 1.0, 1.0, 1.0), (0.9, 1.0, 0.0, 1.0))
    b.add(box, _v3(0, half - F(0.02 * extent), 0), quat_lookat(K_FORWARDS[YM], K_UPS[YM]), _v3(0.4 * extent, 0.4 * extent, 0.01 * extent), lightMat)
    room_tris = (6 * wt * wt + 1) * 12
    mats = []
    for _ in range(n_materials):
        alb = rng.uniform(0.2, 0.9, 3)
        rough = 1.0 if diffuse_only else rng.uniform(0.3, 1.0)
        mats.append(b.material((alb[0], alb[1], alb[2], 1.0), (rough, 1.0, 0.0, 0.0)))
    mats = np.array(mats, np.int32)
    remaining = max(0, target_tris - room_tris)
    n_sph = int(remaining * sphere_fraction) // sph_tris
    n_box = (remaining - n_sph * sph_tris) // 12
    inner = extent * 0.5 * 0.9
    if n_sph:
        T = _keep_clear(rng.uniform(-inner, inner, (n_sph, 3)).astype(F), extent)
        S = np.repeat(rng.uniform(0.004, 0.012, (n_sph, 1)) * extent, 3, 1).astype(F)
        P4, N4, U = _batch_flatten(sph, T, _random_quats(rng, n_sph), S)
        b.add_arrays(P4, N4, U, np.repeat(mats[rng.integers(0, n_materials, n_sph)], sph[0].shape[0]))
    if n_box:
        T = _keep_clear(rng.uniform(-inner, inner, (n_box, 3)).astype(F), extent)
        S = (rng.uniform(0.002, 0.012, (n_box, 3)) * extent).astype(F)
        P4, N4, U = _batch_flatten(box, T, _random_quats(rng, n_box), S)
        b.add_arrays(P4, N4, U, np.repeat(mats[rng.integers(0, n_materials, n_box)], 36))
    return b.build()


def random_rays(scene, n, seed=0, kind="inside"):
    """test rays: origins inside the scene bounds, uniform directions"""
    rng = np.random.default_rng(seed)
    lo = scene.positions[:, :3].min(0) if scene.positions.size else np.array([-1, -1, -1], F)
    hi = scene.positions[:, :3].max(0) if scene.positions.size else np.array([1, 1, 1], F)
    c, e = (lo + hi) * 0.5, (hi - lo) * 0.5
    ro = np.zeros((n, 4), F); rd = np.zeros((n, 4), F)
    ro[:, :3] = c + rng.uniform(-0.9, 0.9, (n, 3)) * e
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rd[:, :3] = d
    return ro, rd


def surface_rays(scene, n, seed=0, offset=1e-4):
    """incoherent secondary rays (SURVEY 8d microbenchmark): origins uniformly on random triangles (+offset*N),
    directions uniform in the hemisphere of the geometric normal, in shuffled order"""
    rng = np.random.default_rng(seed)
    tri = rng.integers(0, scene.tri_count, n)
    P = scene.positions[:, :3].reshape(-1, 3, 3)[tri]
    w = rng.dirichlet([1, 1, 1], n).astype(F)
    p = (P * w[:, :, None]).sum(1)
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where((d * nrm).sum(1, keepdims=True) < 0, -d, d)
    ro = np.zeros((n, 4), F); rd = np.zeros((n, 4), F)
    ro[:, :3] = p + nrm * offset; rd[:, :3] = d
    return ro, rd


def lightmap_pages_for(scene, lm_size, texels_per_meter, max_pages=None, tri_indices=None):
    """Synthetic bake inputs (the atlas packer is out of scope): every triangle gets
    its own chart rasterised at texels_per_meter into lm_size^2 pages in triangle order; covered texels get
    position / normal / sampleCounts = 1, like EmbedTaskFn's outputs (rendering/lightmap.c:1028-1030).
    tri_indices: only these triangles, in this order (a bounded sample of a big scene).
    Returns a list of float32 pages in the reference layout [5][S*S]x4 | [S*S]x3 | [S*S]x3 | [S*S]."""
    S = lm_size
    L = S * S
    pages = []

    def new_page():
        p = np.zeros(L * 27, F)
        pages.append(p)
        return p

    page = new_page()
    cx = cy = rowh = 0
    P = scene.positions[:, :3].reshape(-1, 3, 3).astype(np.float64)
    Nn = scene.normals[:, :3].reshape(-1, 3, 3).astype(np.float64)
    for t in (range(P.shape[0]) if tri_indices is None else tri_indices):
        a, b_, c = P[t]
        e1, e2 = b_ - a, c - a
        l1 = np.linalg.norm(e1)
        if l1 < 1e-9:
            continue
        ux = e1 / l1
        n = np.cross(e1, e2); ln = np.linalg.norm(n)
        if ln < 1e-12:
            continue
        uy = np.cross(n / ln, ux)
        q = np.array([[0, 0], [l1, 0], [e2 @ ux, e2 @ uy]]) * texels_per_meter
        mn = q.min(0); q = q - mn
        w = min(S, int(math.ceil(q[:, 0].max())) + 1); h = min(S, int(math.ceil(q[:, 1].max())) + 1)
        if cx + w > S:
            cx = 0; cy += rowh; rowh = 0
        if cy + h > S:
            if max_pages is not None and len(pages) >= max_pages:
                break
            page = new_page(); cx = cy = rowh = 0
        ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
        d = (q[1, 1] - q[2, 1]) * (q[0, 0] - q[2, 0]) + (q[2, 0] - q[1, 0]) * (q[0, 1] - q[2, 1])
        if abs(d) < 1e-12:
            continue
        w0 = ((q[1, 1] - q[2, 1]) * (xs - q[2, 0]) + (q[2, 0] - q[1, 0]) * (ys - q[2, 1])) / d
        w1 = ((q[2, 1] - q[0, 1]) * (xs - q[2, 0]) + (q[0, 0] - q[2, 0]) * (ys - q[2, 1])) / d
        w2 = 1 - w0 - w1
        inside = (w0 >= -0.05) & (w1 >= -0.05) & (w2 >= -0.05)
        w0c, w1c, w2c = np.clip(w0, 0, 1), np.clip(w1, 0, 1), np.clip(w2, 0, 1)
        s = w0c + w1c + w2c
        w0c, w1c, w2c = w0c / s, w1c / s, w2c / s
        pos = w0c[..., None] * a + w1c[..., None] * b_ + w2c[..., None] * c
        nor = w0c[..., None] * Nn[t, 0] + w1c[..., None] * Nn[t, 1] + w2c[..., None] * Nn[t, 2]
        iy, ix = np.nonzero(inside)
        tex = (cy + iy) * S + (cx + ix)
        posv = page[L * 20:L * 23].reshape(L, 3); norv = page[L * 23:L * 26].reshape(L, 3); cnt = page[L * 26:]
        posv[tex] = pos[iy, ix]; norv[tex] = nor[iy, ix]; cnt[tex] = 1.0
        cx += w; rowh = max(rowh, h)
    return pages
