/*
 * TEST INFRASTRUCTURE -- golden-fixture generator for the lightmap checkpoint format, built from the REFERENCE's own
 * sources where they lie (assets/crate.c, common/guid.c, common/fnv1a.c, io/fstr.c; recipe: oracle/Makefile, target
 * `ref`; output only under oracle/_ref/).  Runs in the build container only; what travels is its output,
 * tests/golden/ref_lmpack.crate, a data file.
 *
 * rendering/lightmap.c itself cannot be compiled here (it includes the Vulkan headers), so the sequence of
 * Crate_Set calls of LmPack_Save (lightmap.c:1225-1262) is restated below around the reference's Crate_* / Guid_*
 * functions; DiskLmPack is restated from lightmap.h:46-54.  After the save the generator adds and overwrites blobs, so
 * that the fixture also pins the reference's slot allocator (best fit, split at >= 1024 spare bytes, overwrite in place).
 *
 *   ref_crate <out.crate>
 */
#include "assets/crate.h"
#include "common/guid.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct DiskLmPack_s { i32 version, directions, lmCount, lmSize, bytesPerLightmap; float texelsPerMeter; } DiskLmPack;
#define kLmPackVersion 2
#define kGiDirections 5

/* the page contents of the fixture: float k of page i (tests/test_host_crate.py builds the same pages) */
static float page_value(int page, int k) { return (float)(page * 1000 + k) * 0.25f - 7.0f; }

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_crate <out.crate>\n"); return 2; }
    remove(argv[1]);
    Crate crate;
    if (!Crate_Open(&crate, argv[1])) { fprintf(stderr, "Crate_Open failed\n"); return 1; }

    const i32 lmcount = 2, lmsize = 4, texelcount = lmsize * lmsize;
    const i32 texelBytes = (i32)(sizeof(float) * 4 * texelcount * kGiDirections + sizeof(float) * 3 * texelcount * 2 + sizeof(float) * texelcount);
    DiskLmPack dpack = { 0 };
    dpack.version = kLmPackVersion; dpack.directions = kGiDirections; dpack.lmCount = lmcount; dpack.lmSize = lmsize;
    dpack.bytesPerLightmap = texelBytes; dpack.texelsPerMeter = 16.0f;
    bool wrote = Crate_Set(&crate, Guid_HashStr("lmpack"), &dpack, sizeof(dpack));
    float* page = malloc((size_t)texelBytes);
    for (i32 i = 0; i < lmcount; ++i)
    {
        char name[64];
        snprintf(name, sizeof(name), "lightmap_%d", i);
        for (i32 k = 0; k < texelBytes / 4; ++k) page[k] = page_value(i, k);
        wrote &= Crate_Set(&crate, Guid_HashStr(name), page, texelBytes);
    }
    /* allocator behaviour the reference handles: further blobs from the tail (best fit, split at >= 1024 spare bytes), an
     * overwrite in place with a smaller size (the slot keeps its size), a second save of the pack over the first.
     * NOT exercised: Crate_Rm / growing a blob.  crate_free (crate.c:98-143) zeroes the freed slot's offset, so the next
     * blob placed there is written at file offset 0, over the header -- a defect, not a format. */
    unsigned char extra[2000];
    for (int k = 0; k < 2000; ++k) extra[k] = (unsigned char)(k * 7 + 3);
    wrote &= Crate_Set(&crate, Guid_HashStr("extra"), extra, 100);
    wrote &= Crate_Set(&crate, Guid_HashStr("small"), extra + 100, 50);
    wrote &= Crate_Set(&crate, Guid_HashStr("extra"), extra + 500, 80);
    for (i32 k = 0; k < texelBytes / 4; ++k) page[k] = page_value(1, k) + 0.5f;
    wrote &= Crate_Set(&crate, Guid_HashStr("lightmap_1"), page, texelBytes);
    i32 off = 0, size = 0;
    wrote &= Crate_Stat(&crate, Guid_HashStr("small"), &off, &size);
    printf("small: offset %d size %d\n", off, size);
    wrote &= Crate_Close(&crate);
    free(page);
    if (!wrote) { fprintf(stderr, "a crate call failed\n"); return 1; }
    return 0;
}
