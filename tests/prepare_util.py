"""a small draw table (Entities columns + mesh table, rendering/drawable.h:14-27) shared by the flatten tests"""
import ctypes

import numpy as np

from pim_amd import api, scenes

C = ctypes


def mesh4(mesh, length=None):
    """box/sphere mesh of scenes.py -> the float4 arrays of Mesh (rendering/mesh.h:25-33)"""
    pos, nor, uv = mesh
    n = pos.shape[0] if length is None else length
    P = np.zeros((n, 4), np.float32); P[:, :3] = pos[:n]; P[:, 3] = 1.0
    N = np.zeros((n, 4), np.float32); N[:, :3] = nor[:n]
    U = np.zeros((n, 4), np.float32); U[:, :2] = uv[:n]; U[:, 2:] = 7.0          # zw of the uv float4 are not copied
    return P, N, U


def draw_table(count=23, seed=5, with_holes=True):
    """`count` entities over four meshes: a box, a small sphere, a mesh whose length is not a multiple of 3
    (its tail keeps zero uvs, path_tracer.c:755) and an empty mesh; some entities have no mesh at all (:706)"""
    rng = np.random.default_rng(seed)
    meshes = [mesh4(scenes.box_mesh()), mesh4(scenes.sphere_mesh(6)), mesh4(scenes.box_mesh(), 34),
              (np.zeros((0, 4), np.float32),) * 3]
    ids = rng.integers(0, len(meshes), count).astype(np.int32)
    if with_holes:
        ids[rng.random(count) < 0.2] = -1
    mats, M = [], np.zeros((count, 4, 4), np.float32)
    for i in range(count):
        t = rng.uniform(-30, 30, 3).astype(np.float32)
        q = scenes.quat_lookat(scenes.normalize3(rng.normal(size=3).astype(np.float32)), np.array([0, 1, 0], np.float32))
        s = rng.uniform(0.2, 3.0, 3).astype(np.float32)
        c0, c1, c2, c3 = scenes.trs_matrix(t, q, s)
        M[i, 0, :3], M[i, 1, :3], M[i, 2, :3], M[i, 3, :3] = c0, c1, c2, c3
        M[i, 3, 3] = 1.0
        m = api.PimMaterial()
        m.albedo, m.rome, m.normal, m.flags, m.ior, m.bumpiness = i, i + 100, -1, i & 3, 1.0 + 0.01 * i, 0.5
        mats.append(m)
    return api.Drawables(meshes, ids, M, mats)


def oracle_flatten(lib, dr, V, Mc):
    P, N = np.zeros((V, 4), np.float32), np.zeros((V, 4), np.float32)
    U, ids = np.zeros((V, 2), np.float32), np.zeros(V, np.int32)
    mats = (api.PimMaterial * max(1, Mc))()
    lib.orc_flatten.restype = None
    lib.orc_flatten(C.byref(dr.desc), P.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), U.ctypes.data_as(C.c_void_p),
                    ids.ctypes.data_as(C.c_void_p), mats)
    return P, N, U, ids, mats


def expected_counts(dr):
    V = sum(dr.table[m].length for m in dr.mesh_ids if m >= 0)
    return V, int((dr.mesh_ids >= 0).sum())


def oracle_brdf_lut(lib, size_x, size_y, passes, seed):
    t = np.zeros((size_y, size_x, 2), np.float32)
    lib.orc_brdf_lut_bake.restype = None
    for prev, num in passes:
        lib.orc_brdf_lut_bake(t.ctypes.data_as(C.c_void_p), C.c_int32(size_x), C.c_int32(size_y), C.c_int32(prev), C.c_int32(num), C.c_uint32(seed))
    return t


# ---------------------------------------------------------------------------------------------
# lightmap embedding (EmbedAttributes, rendering/lightmap.c:941-1045)

def embed_table(lm_size, page_ids, tris_per_mesh=(60, 200), seed=11, chart_texels=(2.0, 14.0), hard_cases=True):
    """meshes as chartnodes_assign (lightmap.c:870-939) leaves them: every triangle carries lightmap coordinates in
    uvs.zw and its page's texture id in texIndices.w.  Charts are random small triangles scattered over the atlas, so
    many overlap or touch (nearest-wins, first-wins on ties); hard_cases adds exact duplicates, zero-length edges, a
    triangle hanging over the page border, one covering a fifth of the page, ids of no page, a sky drawable, an entity
    without a mesh, a mesh without texIndices and a mesh drawn by two entities."""
    rng = np.random.default_rng(seed)
    meshes = []
    for n_tri in tris_per_mesh:
        n = n_tri * 3
        P = np.ones((n, 4), np.float32); P[:, :3] = rng.uniform(-40, 40, (n, 3))
        N = np.zeros((n, 4), np.float32); N[:, :3] = rng.normal(size=(n, 3))
        N[:, :3] /= np.linalg.norm(N[:, :3], axis=1, keepdims=True)
        U = np.zeros((n, 4), np.float32); U[:, :2] = rng.random((n, 2))
        T = np.zeros((n, 4), np.int32); T[:, :3] = rng.integers(0, 50, (n, 3))
        centre = rng.random((n_tri, 1, 2))
        size = rng.uniform(chart_texels[0], chart_texels[1], (n_tri, 1, 1)) / lm_size
        U[:, 2:] = (centre + rng.uniform(-1, 1, (n_tri, 3, 2)) * size).reshape(n, 2)
        T[:, 3] = np.repeat(rng.choice(np.asarray(page_ids, np.int32), n_tri), 3)
        if hard_cases and n_tri >= 12:
            U[3:6, 2:] = U[0:3, 2:]; T[3:6, 3] = T[0, 3]                    # exact duplicate of triangle 0: the first wins
            U[7, 2:] = U[6, 2:]                                              # zero-length edge
            U[9:12, 2:] = [[-0.05, 0.4], [0.1, 0.45], [0.02, 0.6]]           # over the border
            U[12:15, 2:] = [[0.2, 0.2], [0.7, 0.25], [0.4, 0.8]]             # large
            T[15:18, 3] = 12345                                              # no such page
            T[19, 3] = 12345                                                 # only the triangle's FIRST vertex is read (:992)
        meshes.append((P, N, U, T))
    plain = mesh4(scenes.box_mesh())                                         # no texIndices: never mapped
    meshes.append(plain)
    ids = [0, 1, -1, 2, 0, 1] if hard_cases else list(range(len(tris_per_mesh)))
    mats = []
    for i in range(len(ids)):
        m = api.PimMaterial()
        m.albedo, m.rome, m.normal, m.flags, m.ior, m.bumpiness = i, i, -1, 0, 1.0, 0.5
        mats.append(m)
    if hard_cases:
        mats[5].flags = api.PIM_MATFLAG_SKY if hasattr(api, "PIM_MATFLAG_SKY") else 2   # kUnmappedMaterials: skipped
    M = np.zeros((len(ids), 4, 4), np.float32)
    M[:, 0, 0] = M[:, 1, 1] = M[:, 2, 2] = M[:, 3, 3] = 1.0                  # EmbedTaskFn reads mesh space: matrices unused
    return api.Drawables(meshes, np.asarray(ids, np.int32), M, mats)


def new_pages(count, lm_size, fill=0.0):
    return [np.full(lm_size * lm_size * 27, fill, np.float32) for _ in range(count)]


def oracle_embed(lib, dr, pages, lm_size, page_tex_ids=None, threads=8):
    arr = api._page_array(pages, lm_size)
    ids = None if page_tex_ids is None else np.ascontiguousarray(page_tex_ids, np.int32)
    lib.orc_lightmap_embed.restype = None
    lib.orc_lightmap_embed(C.byref(dr.desc), None if ids is None else ids.ctypes.data_as(C.c_void_p), arr,
                           C.c_int32(len(pages)), C.c_int32(lm_size), C.c_int(threads))
    return pages


def page_parts(page, lm_size):
    n = lm_size * lm_size
    return page[:n * 20], page[n * 20:n * 23].reshape(n, 3), page[n * 23:n * 26].reshape(n, 3), page[n * 26:]


def world_mesh_with_charts(sc, lm_size, texels_per_meter, max_pages=2):
    """a stand-in for the atlas packer (LmPack_Pack's chart grouping and atlas search are out of scope): every triangle
    of the flattened scene becomes its own chart, projected on its plane at texels_per_meter and shelf-packed with a
    3-texel gutter into lm_size^2 pages.  Returns (Drawables with ONE world-space mesh and an identity matrix -- the
    lightmapped meshes of the reference are world-space -- carrying uvs.zw / texIndices.w, page count)."""
    S = lm_size
    P = sc.positions[:, :3].reshape(-1, 3, 3).astype(np.float64)
    n = sc.positions.shape[0]
    U = np.zeros((n, 4), np.float32); T = np.full((n, 4), -1, np.int32)
    cx = cy = rowh = 3; page = 0
    for t in range(P.shape[0]):
        a, b, c = P[t]
        e1, e2 = b - a, c - a
        l1 = np.linalg.norm(e1); nrm = np.cross(e1, e2); ln = np.linalg.norm(nrm)
        if l1 < 1e-9 or ln < 1e-12:
            continue
        ux = e1 / l1; uy = np.cross(nrm / ln, ux)
        q = np.array([[0, 0], [l1, 0], [e2 @ ux, e2 @ uy]]) * texels_per_meter
        q = q - q.min(0)
        w, h = int(np.ceil(q[:, 0].max())) + 3, int(np.ceil(q[:, 1].max())) + 3
        if w > S - 6 or h > S - 6:
            continue
        if cx + w > S - 3:
            cx = 3; cy += rowh; rowh = 0
        if cy + h > S - 3:
            page += 1; cx = cy = 3; rowh = 0
            if page >= max_pages:
                break
        U[t * 3:t * 3 + 3, 2:] = (q + [cx, cy]) / S
        T[t * 3:t * 3 + 3, 3] = page
        cx += w; rowh = max(rowh, h)
    Pm = np.ascontiguousarray(sc.positions, np.float32); Nm = np.ascontiguousarray(sc.normals, np.float32)
    m = api.PimMaterial(); m.flags = 0
    M = np.zeros((1, 4, 4), np.float32); M[0, 0, 0] = M[0, 1, 1] = M[0, 2, 2] = M[0, 3, 3] = 1.0
    return api.Drawables([(Pm, Nm, U, T)], np.zeros(1, np.int32), M, [m]), min(page + 1, max_pages)


# ---------------------------------------------------------------------------------------------
# LmPack_Pack's data-parallel steps (chartnodes_create, ChartMaskFn)

def oracle_chartnodes(lib, dr, tpu):
    lib.orc_lightmap_chartnodes.restype = C.c_int32
    n = lib.orc_lightmap_chartnodes(C.byref(dr.desc), C.c_float(tpu), None)
    nodes = np.zeros(n, api.CHARTNODE_DTYPE)
    assert lib.orc_lightmap_chartnodes(C.byref(dr.desc), C.c_float(tpu), nodes.ctypes.data_as(C.c_void_p)) == n
    return nodes


def chart_partition(n_nodes, seed=3, sizes=(1, 2, 5, 40)):
    """charts as contiguous runs of nodes (the reference groups by plane; any grouping exercises the same arithmetic)"""
    rng = np.random.default_rng(seed)
    first = [0]
    while first[-1] < n_nodes:
        first.append(min(n_nodes, first[-1] + int(rng.choice(sizes))))
    return np.asarray(first, np.int32)


def oracle_chart_masks(lib, nodes, first):
    nc = first.shape[0] - 1
    areas, sizes = np.zeros(nc, np.float32), np.zeros((nc, 2), np.int32)
    lib.orc_lightmap_chart_layout.restype = None; lib.orc_lightmap_chart_masks.restype = None
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_lightmap_chart_layout(p(nodes), p(first), C.c_int32(nc), p(areas), p(sizes))
    lens = sizes[:, 0].astype(np.int64) * sizes[:, 1]
    offsets = np.zeros(nc, np.int64); offsets[1:] = np.cumsum(lens)[:-1]
    masks = np.full(int(lens.sum()), 7, np.uint8)
    lib.orc_lightmap_chart_masks(p(nodes), p(first), C.c_int32(nc), p(sizes), p(offsets), p(masks))
    return areas, sizes, offsets, masks
