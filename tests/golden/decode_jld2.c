/* decode_jld2.c -- one-off fixture decoder (TEST INFRASTRUCTURE, not product code).
 *
 * Reads the reference's own test fixture test/assets/symmetricblockexamples.jld2
 * (JLD2 = HDF5 container) and writes its *data* (no code) as flat little-endian
 * binaries that tests/golden/make_golden.py packs into .npz files.
 *
 * Fixture layout (loaded by the reference at
 * test/test_symmetricblockmatrix.jl:9-16):
 *   /blockdict : compound{kvvec: ref} -> 1-D array of refs -> each a scalar
 *   compound{first: vlen UTF-8 key, second: {5 refs}} ; the five refs are
 *   (diagonalblocks, selfindices, offblocks, testindices, trialindices);
 *   each is a 1-D array of refs to leaf datasets: 2-D compound{re,im:f64}
 *   (HDF5 dims reversed w.r.t. Julia's column-major (nrows,ncols)) or 1-D int64.
 *
 * Output (per key, file <outdir>/<key>.bin), all int64 / float64 LE:
 *   magic "BSMFIX01" (8 bytes)
 *   for field in 1..5: int64 count; then per element:
 *     matrix field : int64 nrows, int64 ncols, nrows*ncols*(re,im) column-major
 *     index field  : int64 len, len*int64 (1-based, as stored)
 *
 * Build: gcc -O1 -I/opt/conda/include decode_jld2.c -L/opt/conda/lib -lhdf5 \
 *            -Wl,-rpath,/opt/conda/lib -o decode_jld2
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char *first; hobj_ref_t second[5]; } pair_t;

static void die(const char *m) { fprintf(stderr, "decode_jld2: %s\n", m); exit(1); }

static void write_leaf(hid_t file, hobj_ref_t ref, FILE *out, int expect_matrix) {
    hid_t ds = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, &ref);
    if (ds < 0) die("deref leaf");
    hid_t sp = H5Dget_space(ds);
    int nd = H5Sget_simple_extent_ndims(sp);
    hsize_t dims[4] = {0, 0, 0, 0};
    H5Sget_simple_extent_dims(sp, dims, NULL);
    hid_t ty = H5Dget_type(ds);
    if (expect_matrix) {
        if (nd != 2 || H5Tget_class(ty) != H5T_COMPOUND) die("matrix leaf: unexpected shape/type");
        /* HDF5 dims = (ncols, nrows) for a Julia (nrows, ncols) column-major array */
        int64_t nrows = (int64_t)dims[1], ncols = (int64_t)dims[0];
        hid_t mt = H5Tcreate(H5T_COMPOUND, 16);
        H5Tinsert(mt, "re", 0, H5T_NATIVE_DOUBLE);
        H5Tinsert(mt, "im", 8, H5T_NATIVE_DOUBLE);
        double *buf = malloc((size_t)nrows * ncols * 16 + 16);
        if (H5Dread(ds, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) die("read matrix");
        fwrite(&nrows, 8, 1, out);
        fwrite(&ncols, 8, 1, out);
        fwrite(buf, 16, (size_t)nrows * ncols, out);
        free(buf);
        H5Tclose(mt);
    } else {
        if (nd != 1 || H5Tget_class(ty) != H5T_INTEGER) die("index leaf: unexpected shape/type");
        int64_t len = (int64_t)dims[0];
        int64_t *buf = malloc((size_t)len * 8 + 8);
        if (H5Dread(ds, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) die("read index");
        fwrite(&len, 8, 1, out);
        fwrite(buf, 8, (size_t)len, out);
        free(buf);
    }
    H5Tclose(ty);
    H5Sclose(sp);
    H5Dclose(ds);
}

static void write_field(hid_t file, hobj_ref_t ref, FILE *out, int is_matrix) {
    hid_t ds = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, &ref);
    if (ds < 0) die("deref field");
    hid_t sp = H5Dget_space(ds);
    hsize_t n = 0;
    if (H5Sget_simple_extent_ndims(sp) != 1) die("field not 1-D");
    H5Sget_simple_extent_dims(sp, &n, NULL);
    hobj_ref_t *refs = malloc(sizeof(hobj_ref_t) * (n + 1));
    if (H5Dread(ds, H5T_STD_REF_OBJ, H5S_ALL, H5S_ALL, H5P_DEFAULT, refs) < 0) die("read field refs");
    int64_t cnt = (int64_t)n;
    fwrite(&cnt, 8, 1, out);
    for (hsize_t i = 0; i < n; i++) write_leaf(file, refs[i], out, is_matrix);
    free(refs);
    H5Sclose(sp);
    H5Dclose(ds);
}

int main(int argc, char **argv) {
    if (argc != 3) die("usage: decode_jld2 <fixture.jld2> <outdir>");
    hid_t file = H5Fopen(argv[1], H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file < 0) die("open");
    hid_t ds = H5Dopen2(file, "/blockdict", H5P_DEFAULT);
    if (ds < 0) die("open /blockdict");
    hobj_ref_t kv;
    hid_t t1 = H5Tcreate(H5T_COMPOUND, sizeof(hobj_ref_t));
    H5Tinsert(t1, "kvvec", 0, H5T_STD_REF_OBJ);
    if (H5Dread(ds, t1, H5S_ALL, H5S_ALL, H5P_DEFAULT, &kv) < 0) die("read kvvec");
    H5Dclose(ds);

    hid_t kvd = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, &kv);
    hid_t sp = H5Dget_space(kvd);
    hsize_t npairs = 0;
    H5Sget_simple_extent_dims(sp, &npairs, NULL);
    hobj_ref_t *prefs = malloc(sizeof(hobj_ref_t) * (npairs + 1));
    if (H5Dread(kvd, H5T_STD_REF_OBJ, H5S_ALL, H5S_ALL, H5P_DEFAULT, prefs) < 0) die("read pair refs");

    hid_t strt = H5Tcopy(H5T_C_S1);
    H5Tset_size(strt, H5T_VARIABLE);
    H5Tset_cset(strt, H5T_CSET_UTF8);
    hid_t tup = H5Tcreate(H5T_COMPOUND, 5 * sizeof(hobj_ref_t));
    const char *names[5] = {"1", "2", "3", "4", "5"};
    for (int k = 0; k < 5; k++) H5Tinsert(tup, names[k], k * sizeof(hobj_ref_t), H5T_STD_REF_OBJ);
    hid_t pt = H5Tcreate(H5T_COMPOUND, sizeof(pair_t));
    H5Tinsert(pt, "first", offsetof(pair_t, first), strt);
    H5Tinsert(pt, "second", offsetof(pair_t, second), tup);

    for (hsize_t p = 0; p < npairs; p++) {
        hid_t pd = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, &prefs[p]);
        pair_t pr;
        memset(&pr, 0, sizeof pr);
        if (H5Dread(pd, pt, H5S_ALL, H5S_ALL, H5P_DEFAULT, &pr) < 0) die("read pair");
        char path[4096];
        snprintf(path, sizeof path, "%s/%s.bin", argv[2], pr.first);
        FILE *out = fopen(path, "wb");
        if (!out) die("open output");
        fwrite("BSMFIX01", 1, 8, out);
        /* (diagonalblocks, selfindices, offblocks, testindices, trialindices) */
        const int is_matrix[5] = {1, 0, 1, 0, 0};
        for (int k = 0; k < 5; k++) write_field(file, pr.second[k], out, is_matrix[k]);
        fclose(out);
        fprintf(stderr, "decode_jld2: wrote %s\n", path);
        H5Dclose(pd);
    }
    return 0;
}
