/*
 * pim_crate.c -- the lightmap checkpoint: LmPack_Save / LmPack_Load (rendering/lightmap.c:1225-1308) and the blob container
 * they write to, Crate (assets/crate.{h,c}), with the reference's names, signatures and ON-DISK BYTES; Guid_HashStr
 * (common/guid.c:91-101) over the upper-casing Fnv64 of common/fnv1a.c:3-11,79-127; Lightmap_Upload (lightmap.c:135-144).
 * Plain host C: a checkpoint is file IO, nothing here touches the GPU.
 *
 * File layout (crate.c:7-11,147-194): Guid ids[1024] | i32 offsets[1024] | i32 sizes[1024] (24 576 bytes), then the blobs.
 * The table is a list of extents in file order: slot i covers [offsets[i], offsets[i] + sizes[i]); a null id marks a free
 * extent, sizes[i] == 0 ends the list.  A fresh crate is one free extent of 2^30 bytes behind the header.
 * Pinned byte for byte by tests/golden/ref_lmpack.crate, a file the reference's own crate.c / guid.c / fnv1a.c wrote
 * (tests/test_host_crate.py says how).
 *
 * Two defects of crate.c are NOT reproduced (neither is reached by LmPack_Save into a fresh crate or over a pack of the
 * same size, which is all the fixture and the reference's own callers do):
 *   - crate_free zeroes the freed extent's offset (crate.c:108), so the next blob placed there lands on file offset 0,
 *     over the header; here a free extent keeps its offset;
 *   - crate_allocate opens a table entry for the split-off remainder with a forward copy loop (crate.c:82-87) that smears
 *     entry `next` over every later one; here the entries behind it move down by one.
 * The container's bookkeeping is this layer's own design (one ordered extent list with claim / release, below); only the
 * entry points' names and signatures and the bytes in the file are the reference's.
 */
#include "pim_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(Guid) == 16 && _Alignof(Guid) == 16, "Guid (common/guid.h:7-12)");
_Static_assert(sizeof(DiskLmPack) == 24, "DiskLmPack (lightmap.h:46-54)");

enum { kHeaderSize = (int)((sizeof(Guid) + sizeof(i32) + sizeof(i32)) * kCrateLen) };
#define kTableBytes (sizeof(((Crate*)0)->ids) + sizeof(((Crate*)0)->offsets) + sizeof(((Crate*)0)->sizes))
_Static_assert(kTableBytes == kHeaderSize, "the three tables are contiguous in Crate (crate.c:165 reads them in one call)");

/* ---- Guid, common/guid.c + common/fnv1a.c ---- */

#define Fnv64Bias  14695981039346656037ull
#define Fnv64Prime 1099511628211ull

static char FnvToUpper(char x)                      /* fnv1a.c:3-11: branch-free ASCII upper-casing */
{
    i32 c = (i32)x;
    i32 lo = ('a' - 1) - c;
    i32 hi = c - ('z' + 1);
    i32 mask = (lo & hi) >> 9;
    c = c + (mask & ('A' - 'a'));
    return (char)c;
}

static u64 Fnv64String(const char* ptr, u64 hash)   /* fnv1a.c:79-82,116-124: the char is sign-extended before the xor */
{
    for (i32 i = 0; ptr[i]; ++i) hash = (hash ^ (u64)FnvToUpper(ptr[i])) * Fnv64Prime;
    return hash;
}

bool Guid_IsNull(Guid x) { return !(x.a | x.b); }
bool Guid_Equal(Guid lhs, Guid rhs) { return !((lhs.a - rhs.a) | (lhs.b - rhs.b)); }

Guid Guid_HashStr(const char* str)                  /* guid.c:91-101 */
{
    Guid id = { 0, 0 };
    if (str && str[0])
    {
        id.a = Fnv64String(str, Fnv64Bias);
        id.b = Fnv64String(str, id.a);
        id.a = id.a ? id.a : 1;
    }
    return id;
}

Guid Guid_FromStr(const char* str) { return Guid_HashStr(str); }     /* guid.c:103-111 also records the name for the editor's UI */

/* ---- Crate: the on-disk format is the reference's (assets/crate.c:7-11,147-260); the bookkeeping below is this layer's own ---- */

/*
 * The header's three parallel arrays are ONE ordered list of extents that tiles the file behind the header without gaps:
 * entry k covers [offset, offset + size), a null id marks it free, the first size of 0 ends the list.  Everything the
 * container does is one of four list operations on that view -- look an id up, pick the tightest free extent, cut an
 * extent in two, give an extent back and fuse it with free neighbours -- so they are written once, over an Extent value
 * type, instead of index arithmetic on three arrays at every call site.
 */
typedef struct Extent { Guid id; i32 offset; i32 size; } Extent;

enum { kMinRemainder = 1024 };          /* a placed blob leaves a free extent behind it only if that many bytes remain (crate.c:56-96) */

static Extent ext_at(const Crate* c, i32 k) { Extent e = { c->ids[k], c->offsets[k], c->sizes[k] }; return e; }
static void ext_put(Crate* c, i32 k, Extent e) { c->ids[k] = e.id; c->offsets[k] = e.offset; c->sizes[k] = e.size; }
static bool ext_is_free(Extent e) { return Guid_IsNull(e.id); }

static i32 ext_count(const Crate* c)
{
    i32 n = 0;
    while (n < kCrateLen && c->sizes[n] != 0) ++n;
    return n;
}

/* opens a hole at position k (entries k.. move one place towards the end); false when the table is full */
static bool ext_open_gap(Crate* c, i32 k)
{
    const i32 n = ext_count(c);
    if (n >= kCrateLen || k > n) return false;
    for (i32 j = n; j > k; --j) ext_put(c, j, ext_at(c, j - 1));
    return true;
}

/* closes position k (entries behind it move one place towards the front, the vacated last entry becomes the list's end) */
static void ext_close_gap(Crate* c, i32 k)
{
    const i32 n = ext_count(c);
    for (i32 j = k; j + 1 < n; ++j) ext_put(c, j, ext_at(c, j + 1));
    if (n > 0) ext_put(c, n - 1, (Extent){ { 0, 0 }, 0, 0 });
}

static i32 ext_lookup(const Crate* c, Guid id)
{
    if (Guid_IsNull(id)) return -1;
    const i32 n = ext_count(c);
    for (i32 k = 0; k < n; ++k)
        if (Guid_Equal(c->ids[k], id)) return k;
    return -1;
}

/* the free extent that wastes the least (the first of equally tight ones), -1 if none is large enough */
static i32 ext_tightest_free(const Crate* c, i32 bytes)
{
    i32 pick = -1;
    const i32 n = ext_count(c);
    for (i32 k = 0; k < n; ++k)
    {
        const Extent e = ext_at(c, k);
        if (!ext_is_free(e) || e.size < bytes) continue;
        if (pick < 0 || e.size < c->sizes[pick]) pick = k;
    }
    return pick;
}

/* gives the free extent k to `id`; what it does not need stays behind as a free extent of its own when it is worth one */
static void ext_claim(Crate* c, i32 k, Guid id, i32 bytes)
{
    Extent e = ext_at(c, k);
    const i32 rest = e.size - bytes;
    e.id = id;
    if (rest >= kMinRemainder && ext_open_gap(c, k + 1))
    {
        ext_put(c, k + 1, (Extent){ { 0, 0 }, e.offset + bytes, rest });
        e.size = bytes;
    }
    ext_put(c, k, e);
}

/* returns extent k to the free list and fuses the run of free extents it is now part of into one (its offset is the run's) */
static void ext_release(Crate* c, i32 k)
{
    c->ids[k] = (Guid){ 0, 0 };
    while (k > 0 && ext_is_free(ext_at(c, k - 1))) --k;                 /* first entry of the free run */
    while (k + 1 < ext_count(c) && ext_is_free(ext_at(c, k + 1)))
    {
        c->sizes[k] += c->sizes[k + 1];
        ext_close_gap(c, k + 1);
    }
}

static void* table_bytes(Crate* c) { return c->ids; }                  /* ids | offsets | sizes are contiguous (static assert above) */

static bool io_at(FILE* f, i32 offset, void* rd, const void* wr, i32 bytes)
{
    if (fseek(f, (long)offset, SEEK_SET) != 0) return false;
    const size_t n = (size_t)bytes;
    return rd ? fread(rd, 1, n, f) == n : fwrite(wr, 1, n, f) == n;
}

bool Crate_Open(Crate* crate, const char* path)
{
    if (!crate || !path) return false;
    memset(crate, 0, sizeof(*crate));
    FILE* f = fopen(path, "rb+");
    const bool fresh = f == NULL;
    if (fresh && !(f = fopen(path, "wb+"))) return false;
    bool good;
    if (fresh)
    {
        ext_put(crate, 0, (Extent){ { 0, 0 }, kHeaderSize, 1 << 30 });    /* one free extent behind the header */
        good = io_at(f, 0, NULL, table_bytes(crate), kHeaderSize);
    }
    else good = io_at(f, 0, table_bytes(crate), NULL, kHeaderSize);       /* shorter than a header: not a crate */
    if (!good) { fclose(f); memset(crate, 0, sizeof(*crate)); return false; }
    crate->file.handle = f;
    return true;
}

bool Crate_Close(Crate* crate)
{
    if (!crate) return false;
    FILE* f = crate->file.handle;
    if (f)
    {
        io_at(f, 0, NULL, table_bytes(crate), kHeaderSize);               /* the table goes out last */
        fflush(f);
        fclose(f);
    }
    memset(crate, 0, sizeof(*crate));
    return f != NULL;
}

bool Crate_Get(Crate* crate, Guid id, void* dst, i32 size)
{
    if (!crate || !crate->file.handle || !dst || size <= 0) return false;
    const i32 k = ext_lookup(crate, id);
    return k >= 0 && size <= crate->sizes[k] && io_at(crate->file.handle, crate->offsets[k], dst, NULL, size);
}

bool Crate_Set(Crate* crate, Guid id, const void* src, i32 size)
{
    if (!crate || !crate->file.handle || !src || size <= 0 || Guid_IsNull(id)) return false;
    i32 k = ext_lookup(crate, id);
    if (k >= 0 && crate->sizes[k] < size) { ext_release(crate, k); k = -1; }     /* outgrew its extent: it moves */
    if (k < 0)
    {
        if ((k = ext_tightest_free(crate, size)) < 0) return false;
        ext_claim(crate, k, id, size);
    }
    return io_at(crate->file.handle, crate->offsets[k], NULL, src, size);
}

bool Crate_Rm(Crate* crate, Guid id)
{
    const i32 k = crate ? ext_lookup(crate, id) : -1;
    if (k >= 0) ext_release(crate, k);
    return k >= 0;
}

bool Crate_Stat(const Crate* crate, Guid id, i32* offsetOut, i32* sizeOut)
{
    const i32 k = crate ? ext_lookup(crate, id) : -1;
    if (k < 0) return false;
    if (offsetOut) *offsetOut = crate->offsets[k];
    if (sizeOut) *sizeOut = crate->sizes[k];
    return true;
}

/* ---- the lightmap pack's consumers ---- */

static LmUploadFn ms_uploader;
static void* ms_uploaderUser;
void PtSys_SetLightmapUploader(LmUploadFn fn, void* user) { ms_uploader = fn; ms_uploaderUser = user; }

/* lightmap.c:135-144: layer i of the page's 2D-array texture (kGiDirections layers of size x size RGBA32F) is probes[i],
 * sizeof(float4) * size * size bytes.  vkrTexTable_Upload is the renderer's; the embedding code registers its stand-in. */
void Lightmap_Upload(Lightmap* lm)
{
    if (!lm || !ms_uploader || lm->size <= 0) return;
    LmPack_Sync();                                  /* resident pages (LmPack_SetResident(2)): bring the bake's results to the host first */
    const i32 len = lm->size * lm->size;
    for (i32 i = 0; i < kGiDirections; ++i)
        ms_uploader(lm->slot, i, lm->probes[i], (i32)sizeof(float4) * len, ms_uploaderUser);
}

/*
 * The checkpoint's blobs (rendering/lightmap.c:1225-1308; DiskLmPack lightmap.h:46-54): "lmpack" = the 24-byte header,
 * "lightmap_<i>" = page i as the one block Lightmap_New makes (probes | position | normal | sampleCounts, lightmap.c:95-110).
 */
static Guid lm_blob_id(i32 page)                     /* page < 0: the header */
{
    char name[32];
    if (page < 0) return Guid_FromStr("lmpack");
    snprintf(name, sizeof(name), "lightmap_%d", page);
    return Guid_FromStr(name);
}

static i32 lm_page_bytes(i32 lmSize)
{
    const i32 texels = lmSize * lmSize;
    const i32 perTexel = (i32)sizeof(float4) * kGiDirections + (i32)sizeof(float3) * 2 + (i32)sizeof(float);
    return perTexel * texels;
}

static DiskLmPack lm_header_of(const LmPack* pack)
{
    const DiskLmPack h = {
        .version = kLmPackVersion, .directions = kGiDirections, .lmCount = pack->lmCount, .lmSize = pack->lmSize,
        .bytesPerLightmap = lm_page_bytes(pack->lmSize), .texelsPerMeter = pack->texelsPerMeter,
    };
    return h;
}

static bool lm_header_usable(const DiskLmPack* h)
{
    return h->version == kLmPackVersion && h->directions == kGiDirections && h->lmCount > 0 && h->lmSize > 0;
}

bool LmPack_Save(Crate* crate, const LmPack* pack)
{
    if (!crate || !pack) return false;
    if (pack == LmPack_Get()) LmPack_Sync();         /* resident pages: what is saved is the bake as it stands */
    const DiskLmPack header = lm_header_of(pack);
    if (!Crate_Set(crate, lm_blob_id(-1), &header, (i32)sizeof(header))) return false;
    i32 failed = 0;
    for (i32 i = 0; i < pack->lmCount; ++i)          /* every page is attempted even after a failure, as in the reference */
        if (!Crate_Set(crate, lm_blob_id(i), pack->lightmaps[i].probes[0], header.bytesPerLightmap)) ++failed;
    return failed == 0;
}

bool LmPack_Load(Crate* crate, LmPack* pack)
{
    if (!crate || !pack) return false;
    LmPack_Del(pack);                                /* whatever it held is gone, header or no header */
    DiskLmPack header; memset(&header, 0, sizeof(header));
    if (!Crate_Get(crate, lm_blob_id(-1), &header, (i32)sizeof(header)) || !lm_header_usable(&header)) return false;
    Lightmap* pages = calloc((size_t)header.lmCount, sizeof(Lightmap));
    if (!pages) return false;
    pack->lightmaps = pages;
    pack->lmCount = header.lmCount;
    pack->lmSize = header.lmSize;
    pack->texelsPerMeter = header.texelsPerMeter;
    const i32 pageBytes = lm_page_bytes(header.lmSize);
    i32 failed = 0;
    for (i32 i = 0; i < header.lmCount; ++i)
    {
        Lightmap_New(&pages[i], header.lmSize);
        if (!pages[i].probes[0] || !Crate_Get(crate, lm_blob_id(i), pages[i].probes[0], pageBytes)) { ++failed; continue; }
        Lightmap_Upload(&pages[i]);
    }
    return failed == 0;
}
