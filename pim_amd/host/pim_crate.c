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
 */
#include "pim_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(Guid) == 16 && _Alignof(Guid) == 16, "Guid (common/guid.h:7-12)");
_Static_assert(sizeof(DiskLmPack) == 24, "DiskLmPack (lightmap.h:46-54)");

enum { kSplitSize = 1024, kHeaderSize = (int)((sizeof(Guid) + sizeof(i32) + sizeof(i32)) * kCrateLen) };
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

/* ---- Crate, assets/crate.c ---- */

static i32 crate_find(const Crate* crate, Guid key)                  /* crate.c:15-30 */
{
    if (!Guid_IsNull(key))
        for (i32 i = 0; (i < kCrateLen) && crate->sizes[i]; ++i)
            if (Guid_Equal(crate->ids[i], key)) return i;
    return -1;
}

static i32 crate_findempty(const Crate* crate, i32 size)             /* crate.c:32-54: best fit among the free extents */
{
    i32 chosen = -1, chosenDiff = 1 << 30;
    if (size > 0)
        for (i32 i = 0; (i < kCrateLen) && crate->sizes[i]; ++i)
            if (Guid_IsNull(crate->ids[i]))
            {
                const i32 diff = crate->sizes[i] - size;
                if ((diff >= 0) && (diff < chosenDiff)) { chosen = i; chosenDiff = diff; }
            }
    return chosen;
}

static i32 crate_allocate(Crate* crate, Guid id, i32 size)           /* crate.c:56-96 */
{
    const i32 slot = crate_findempty(crate, size);
    if (slot < 0) return -1;
    crate->ids[slot] = id;
    const i32 diff = crate->sizes[slot] - size;
    if (diff >= kSplitSize)
    {
        const i32 next = slot + 1;
        if ((next < kCrateLen) && Guid_IsNull(crate->ids[kCrateLen - 1]) && !crate->sizes[kCrateLen - 1])
        {
            /* the remainder becomes a free extent of its own right behind the blob */
            memmove(&crate->ids[next + 1], &crate->ids[next], sizeof(Guid) * (size_t)(kCrateLen - 1 - next));
            memmove(&crate->offsets[next + 1], &crate->offsets[next], sizeof(i32) * (size_t)(kCrateLen - 1 - next));
            memmove(&crate->sizes[next + 1], &crate->sizes[next], sizeof(i32) * (size_t)(kCrateLen - 1 - next));
            crate->ids[next] = (Guid){ 0, 0 };
            crate->offsets[next] = crate->offsets[slot] + size;
            crate->sizes[next] = diff;
            crate->sizes[slot] = size;
        }
    }
    return slot;
}

static void crate_remove_entry(Crate* crate, i32 slot)
{
    memmove(&crate->ids[slot], &crate->ids[slot + 1], sizeof(Guid) * (size_t)(kCrateLen - 1 - slot));
    memmove(&crate->offsets[slot], &crate->offsets[slot + 1], sizeof(i32) * (size_t)(kCrateLen - 1 - slot));
    memmove(&crate->sizes[slot], &crate->sizes[slot + 1], sizeof(i32) * (size_t)(kCrateLen - 1 - slot));
    crate->ids[kCrateLen - 1] = (Guid){ 0, 0 };
    crate->offsets[kCrateLen - 1] = 0;
    crate->sizes[kCrateLen - 1] = 0;
}

static void crate_free(Crate* crate, i32 slot)                       /* crate.c:98-143: free + merge with free neighbours */
{
    crate->ids[slot] = (Guid){ 0, 0 };
    const i32 prev = slot - 1;
    if ((prev >= 0) && Guid_IsNull(crate->ids[prev]))
    {
        crate->sizes[prev] += crate->sizes[slot];
        crate_remove_entry(crate, slot);
        --slot;
    }
    const i32 next = slot + 1;
    if ((next < kCrateLen) && crate->sizes[next] && Guid_IsNull(crate->ids[next]))
    {
        crate->sizes[slot] += crate->sizes[next];
        crate_remove_entry(crate, next);
    }
}

static bool file_seek(FILE* f, i32 offset) { return fseek(f, (long)offset, SEEK_SET) == 0; }

bool Crate_Open(Crate* crate, const char* path)                      /* crate.c:147-175 */
{
    if (!crate || !path) return false;
    memset(crate, 0, sizeof(*crate));
    bool exists = true;
    FILE* file = fopen(path, "rb+");
    if (!file) { exists = false; file = fopen(path, "wb+"); }
    if (!file) return false;
    crate->file.handle = file;
    file_seek(file, 0);
    if (exists)
    {
        if (fread(crate->ids, 1, kTableBytes, file) != kTableBytes)
        {
            /* shorter than a header: not a crate (the reference goes on with whatever it read) */
            fclose(file); memset(crate, 0, sizeof(*crate));
            return false;
        }
    }
    else
    {
        crate->offsets[0] = kHeaderSize;
        crate->sizes[0] = 1 << 30;
        fwrite(crate->ids, 1, kTableBytes, file);
    }
    file_seek(file, kHeaderSize);
    return true;
}

bool Crate_Close(Crate* crate)                                       /* crate.c:177-194 */
{
    bool wasopen = false;
    if (crate)
    {
        FILE* file = crate->file.handle;
        if (file)
        {
            file_seek(file, 0);
            fwrite(crate->ids, 1, kTableBytes, file);
            fflush(file);
            fclose(file);
            wasopen = true;
        }
        memset(crate, 0, sizeof(*crate));
    }
    return wasopen;
}

bool Crate_Get(Crate* crate, Guid id, void* dst, i32 size)          /* crate.c:196-219 */
{
    if (!crate || !crate->file.handle || !dst || size <= 0) return false;
    const i32 slot = crate_find(crate, id);
    if (slot < 0 || size > crate->sizes[slot]) return false;
    FILE* file = crate->file.handle;
    if (!file_seek(file, crate->offsets[slot])) return false;
    return fread(dst, 1, (size_t)size, file) == (size_t)size;
}

bool Crate_Set(Crate* crate, Guid id, const void* src, i32 size)    /* crate.c:221-260 */
{
    if (!crate || !crate->file.handle || !src || size <= 0 || Guid_IsNull(id)) return false;
    i32 slot = crate_find(crate, id);
    if (slot >= 0 && crate->sizes[slot] < size) { crate_free(crate, slot); slot = -1; }      /* grown: a new place */
    if (slot < 0) slot = crate_allocate(crate, id, size);
    if (slot < 0) return false;
    FILE* file = crate->file.handle;
    if (!file_seek(file, crate->offsets[slot])) return false;
    return fwrite(src, 1, (size_t)size, file) == (size_t)size;
}

bool Crate_Rm(Crate* crate, Guid id)                                /* crate.c:262-271 */
{
    const i32 slot = crate ? crate_find(crate, id) : -1;
    if (slot < 0) return false;
    crate_free(crate, slot);
    return true;
}

bool Crate_Stat(const Crate* crate, Guid id, i32* offsetOut, i32* sizeOut)      /* crate.c:273-285 */
{
    const i32 slot = crate ? crate_find(crate, id) : -1;
    if (slot < 0) return false;
    if (offsetOut) *offsetOut = crate->offsets[slot];
    if (sizeOut) *sizeOut = crate->sizes[slot];
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

static i32 lm_texel_bytes(i32 lmsize)               /* lightmap.c:1232-1238: probes + position + normal + sampleCounts */
{
    const i32 texelcount = lmsize * lmsize;
    return (i32)sizeof(float4) * texelcount * kGiDirections + (i32)sizeof(float3) * texelcount * 2 + (i32)sizeof(float) * texelcount;
}

bool LmPack_Save(Crate* crate, const LmPack* pack)                   /* lightmap.c:1225-1262 */
{
    if (!crate || !pack) return false;
    if (pack == LmPack_Get()) LmPack_Sync();         /* resident pages: what is saved is the bake as it stands */
    bool wrote = false;
    const i32 lmcount = pack->lmCount;
    const i32 texelBytes = lm_texel_bytes(pack->lmSize);
    DiskLmPack dpack; memset(&dpack, 0, sizeof(dpack));
    dpack.version = kLmPackVersion;
    dpack.directions = kGiDirections;
    dpack.lmCount = lmcount;
    dpack.lmSize = pack->lmSize;
    dpack.bytesPerLightmap = texelBytes;
    dpack.texelsPerMeter = pack->texelsPerMeter;
    if (Crate_Set(crate, Guid_FromStr("lmpack"), &dpack, (i32)sizeof(dpack)))
    {
        wrote = true;
        for (i32 i = 0; i < lmcount; ++i)
        {
            char name[64];
            snprintf(name, sizeof(name), "lightmap_%d", i);
            const Lightmap lm = pack->lightmaps[i];
            wrote &= Crate_Set(crate, Guid_FromStr(name), lm.probes[0], texelBytes);     /* the page is ONE block (lightmap.c:95-110) */
        }
    }
    return wrote;
}

bool LmPack_Load(Crate* crate, LmPack* pack)                         /* lightmap.c:1264-1308 */
{
    if (!crate || !pack) return false;
    bool loaded = false;
    LmPack_Del(pack);
    DiskLmPack dpack; memset(&dpack, 0, sizeof(dpack));
    if (Crate_Get(crate, Guid_FromStr("lmpack"), &dpack, (i32)sizeof(dpack)))
    {
        if ((dpack.version == kLmPackVersion) && (dpack.directions == kGiDirections) && (dpack.lmCount > 0) && (dpack.lmSize > 0))
        {
            const i32 lmcount = dpack.lmCount, lmsize = dpack.lmSize;
            const i32 texelBytes = lm_texel_bytes(lmsize);
            pack->lightmaps = calloc((size_t)lmcount, sizeof(pack->lightmaps[0]));
            if (!pack->lightmaps) return false;
            loaded = true;
            pack->lmCount = lmcount;
            pack->lmSize = lmsize;
            pack->texelsPerMeter = dpack.texelsPerMeter;
            for (i32 i = 0; i < lmcount; ++i)
            {
                char name[64];
                snprintf(name, sizeof(name), "lightmap_%d", i);
                Lightmap lm; memset(&lm, 0, sizeof(lm));
                Lightmap_New(&lm, lmsize);
                if (!lm.probes[0]) { loaded = false; pack->lightmaps[i] = lm; continue; }
                loaded &= Crate_Get(crate, Guid_FromStr(name), lm.probes[0], texelBytes);
                Lightmap_Upload(&lm);
                pack->lightmaps[i] = lm;
            }
        }
    }
    return loaded;
}
