/*
 * rjo_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See rjo_oracle.h for scope and the parity pin.  Every function names the
 * reference lines it restates.  Written from the reference's behaviour, in
 * plain C99, single-threaded (the reference's OpenMP team only changes row
 * order, which is not part of the contract: results are multisets).
 */
#include "rjo_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ cells --
 * `Data = variant<int32_t,int64_t,double,std::string,monostate>`
 * (reference include/statement.h:13).  Tags follow the variant index.        */
enum { T_I32 = 0, T_I64 = 1, T_F64 = 2, T_STR = 3, T_NULL = 4 };

typedef struct cell {
    uint8_t  tag;
    uint32_t len; /* T_STR only */
    union {
        int32_t     i32;
        int64_t     i64;
        double      f64;
        const char* s;
    } v;
} cell;

/* A row store: `std::vector<std::vector<Data>>` flattened row-major. */
typedef struct rows {
    size_t n, w;
    cell*  c;
    /* heap blocks owned by this row store (long strings spanning pages) */
    char** blobs;
    size_t n_blobs, cap_blobs;
} rows;

typedef struct errbuf {
    char*  p;
    size_t cap;
} errbuf;

static int fail(errbuf* e, const char* fmt, ...) {
    if (e && e->p && e->cap) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(e->p, e->cap, fmt, ap);
        va_end(ap);
    }
    return -1;
}

static void rows_free(rows* r) {
    if (!r) return;
    for (size_t i = 0; i < r->n_blobs; ++i) free(r->blobs[i]);
    free(r->blobs);
    free(r->c);
    memset(r, 0, sizeof *r);
}

static int rows_init(rows* r, size_t n, size_t w) {
    memset(r, 0, sizeof *r);
    r->n = n;
    r->w = w;
    size_t cells = n * w;
    r->c = (cell*)malloc((cells ? cells : 1) * sizeof(cell));
    if (!r->c) return -1;
    for (size_t i = 0; i < cells; ++i) {
        r->c[i].tag = T_NULL; /* std::monostate{} default, build_table.cpp:314-315 */
        r->c[i].len = 0;
        r->c[i].v.i64 = 0;
    }
    return 0;
}

static int rows_add_blob(rows* r, char* b) {
    if (r->n_blobs == r->cap_blobs) {
        size_t nc = r->cap_blobs ? r->cap_blobs * 2 : 8;
        char** nb = (char**)realloc(r->blobs, nc * sizeof(char*));
        if (!nb) return -1;
        r->blobs = nb;
        r->cap_blobs = nc;
    }
    r->blobs[r->n_blobs++] = b;
    return 0;
}

/* blobs move with the rows that reference them */
static void rows_take_blobs(rows* dst, rows* src) {
    for (size_t i = 0; i < src->n_blobs; ++i) rows_add_blob(dst, src->blobs[i]);
    src->n_blobs = 0;
}

static inline uint16_t rd16(const uint8_t* p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}

/* get_bitmap — reference build_table.cpp:306-310 */
static inline int get_bit(const uint8_t* bitmap, uint32_t idx) {
    return (bitmap[idx / 8] >> (idx % 8)) & 1u;
}

/* ---------------------------------------------------------- from_columnar --
 * One column of Table::from_columnar, reference build_table.cpp:317-431.
 * Writes into column `ci` of the row store.                                  */
static int decode_column(const rj_column* col, uint64_t num_rows, rows* out, size_t ci, errbuf* e) {
    size_t row_idx = 0;
    for (uint64_t pi = 0; pi < col->n_pages; ++pi) {
        const uint8_t* page = (const uint8_t*)col->pages[pi];
        uint16_t       nr = rd16(page);
        switch (col->type) {
        case RJ_INT32:
        case RJ_INT64:
        case RJ_FP64: {
            /* build_table.cpp:325-381: values dense from +4 (INT32) / +8 */
            size_t         vsz = col->type == RJ_INT32 ? 4 : 8;
            const uint8_t* data = page + (col->type == RJ_INT32 ? 4 : 8);
            const uint8_t* bitmap = page + RJ_PAGE_SIZE - (nr + 7) / 8;
            uint32_t       data_idx = 0;
            for (uint32_t i = 0; i < nr; ++i) {
                if (get_bit(bitmap, i)) {
                    if (row_idx >= num_rows) return fail(e, "row_idx");
                    cell* c = &out->c[row_idx * out->w + ci];
                    if (col->type == RJ_INT32) {
                        c->tag = T_I32;
                        memcpy(&c->v.i32, data + (size_t)data_idx * vsz, 4);
                    } else if (col->type == RJ_INT64) {
                        c->tag = T_I64;
                        memcpy(&c->v.i64, data + (size_t)data_idx * vsz, 8);
                    } else {
                        c->tag = T_F64;
                        memcpy(&c->v.f64, data + (size_t)data_idx * vsz, 8);
                    }
                    ++data_idx;
                    ++row_idx;
                } else {
                    /* the reference does not bounds-check NULL rows (:339-340): a NULL
                     * row past num_rows only advances row_idx (nothing is written), and
                     * "row_idx" is raised only by a non-NULL value at or past the end */
                    ++row_idx;
                }
            }
            break;
        }
        case RJ_VARCHAR: {
            if (nr == 0xffff) {
                /* first page of a long string: one row (build_table.cpp:384-391) */
                uint16_t nchars = rd16(page + 2);
                if (row_idx >= num_rows) return fail(e, "row_idx");
                cell* c = &out->c[row_idx * out->w + ci];
                char* b = (char*)malloc(nchars ? nchars : 1);
                if (!b || rows_add_blob(out, b)) return fail(e, "oom");
                memcpy(b, page + 4, nchars);
                c->tag = T_STR;
                c->len = nchars;
                c->v.s = b;
                ++row_idx;
            } else if (nr == 0xfffe) {
                /* continuation: append to the previous row (build_table.cpp:392-405) */
                uint16_t nchars = rd16(page + 2);
                if (row_idx == 0) return fail(e, "long string page 0xfffe must follows a string");
                cell* c = &out->c[(row_idx - 1) * out->w + ci];
                if (c->tag != T_STR) return fail(e, "long string page 0xfffe must follows a string");
                char* b = (char*)malloc((size_t)c->len + nchars + 1);
                if (!b || rows_add_blob(out, b)) return fail(e, "oom");
                memcpy(b, c->v.s, c->len);
                memcpy(b + c->len, page + 4, nchars);
                c->v.s = b;
                c->len += nchars;
            } else {
                /* normal page (build_table.cpp:406-427): offsets are END offsets
                 * relative to the char area, which starts after n_nonnull u16s */
                uint16_t       nnn = rd16(page + 2);
                const uint8_t* offs = page + 4;
                const char*    data_begin = (const char*)page + 4 + (size_t)nnn * 2;
                const char*    str_begin = data_begin;
                const uint8_t* bitmap = page + RJ_PAGE_SIZE - (nr + 7) / 8;
                uint32_t       data_idx = 0;
                for (uint32_t i = 0; i < nr; ++i) {
                    if (get_bit(bitmap, i)) {
                        /* only a non-NULL value past the end raises (:418-420) */
                        if (row_idx >= num_rows) return fail(e, "row_idx");
                        uint16_t    off = rd16(offs + (size_t)data_idx * 2);
                        cell*       c = &out->c[row_idx * out->w + ci];
                        const char* end = data_begin + off;
                        c->tag = T_STR;
                        c->len = (uint32_t)(end - str_begin);
                        c->v.s = str_begin;
                        str_begin = end;
                        ++data_idx;
                    }
                    ++row_idx;
                }
            }
            break;
        }
        default: return fail(e, "unknown column type %d", col->type);
        }
    }
    return 0;
}

/* Table::from_columnar — reference build_table.cpp:312-436 */
static int from_columnar(const rj_input* in, rows* out, errbuf* e) {
    if (rows_init(out, (size_t)in->num_rows, (size_t)in->n_cols)) return fail(e, "oom");
    for (uint64_t ci = 0; ci < in->n_cols; ++ci)
        if (decode_column(&in->cols[ci], in->num_rows, out, (size_t)ci, e)) return -1;
    return 0;
}

/* ------------------------------------------------------------------ hashes --
 * HashUtil<K>::hash — reference src/execute.cpp:16-41                        */
uint64_t rjo_hash_int(int64_t key) {
    uint64_t k = (uint64_t)key; /* int32 keys are sign-extended first (:21) */
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static uint64_t hash_f64(double d) { /* :28-31, bit pattern through the int hash */
    uint64_t bits;
    memcpy(&bits, &d, 8);
    return rjo_hash_int((int64_t)bits);
}

static uint64_t hash_str(const char* s, uint32_t len) { /* FNV-1a, :33-38 */
    uint64_t h = 14695981039346656037ULL;
    for (uint32_t i = 0; i < len; ++i) {
        /* `static_cast<size_t>(c)` on a (signed) char sign-extends */
        h ^= (uint64_t)(int64_t)(signed char)s[i];
        h *= 1099511628211ULL;
    }
    return h;
}

static uint64_t hash_cell(const cell* c) {
    switch (c->tag) {
    case T_I32: return rjo_hash_int((int64_t)c->v.i32);
    case T_I64: return rjo_hash_int(c->v.i64);
    case T_F64: return hash_f64(c->v.f64);
    case T_STR: return hash_str(c->v.s, c->len);
    default: return 0;
    }
}

/* `slot_key[h] != key` with K's operator!= (src/execute.cpp:215,231) */
static int key_equal(const cell* a, const cell* b) {
    switch (a->tag) {
    case T_I32: return a->v.i32 == b->v.i32;
    case T_I64: return a->v.i64 == b->v.i64;
    case T_F64: return a->v.f64 == b->v.f64; /* NaN never equal; -0.0 == +0.0 */
    case T_STR: return a->len == b->len && memcmp(a->v.s, b->v.s, a->len) == 0;
    default: return 0;
    }
}

/* bucket-count rule — reference src/execute.cpp:86-92, SPC__LEVEL2_CACHE_SIZE
 * = 1048576 (include/hardware.h:44)                                          */
uint64_t rjo_num_buckets(uint64_t B, uint64_t key_bytes) {
    const uint64_t L2 = 1048576;
    uint64_t       approx = (B * (key_bytes + 4) + L2 - 1) / L2;
    if (approx < 1) approx = 1;
    if (approx > 128) approx = 128;
    uint64_t nb = 1;
    while (nb < approx) nb <<= 1;
    return nb;
}

/* ---------------------------------------------------------------- executor */
static int exec_node(const rj_plan* plan, uint64_t node_idx, rows* out, errbuf* e);

/* execute_scan — reference src/execute.cpp:284-300 */
static int exec_scan(const rj_plan* plan, const rj_node* n, rows* out, errbuf* e) {
    if (n->base_table_id >= plan->n_inputs) return fail(e, "scan: bad base_table_id");
    const rj_input* in = &plan->inputs[n->base_table_id];
    rows            table;
    if (from_columnar(in, &table, e)) {
        rows_free(&table);
        return -1;
    }
    if (rows_init(out, table.n, (size_t)n->n_out)) {
        rows_free(&table);
        return fail(e, "oom");
    }
    for (uint64_t k = 0; k < n->n_out; ++k)
        if (n->out_idx[k] >= table.w) {
            rows_free(&table);
            return fail(e, "scan: output attr out of range");
        }
    for (size_t r = 0; r < table.n; ++r)
        for (uint64_t k = 0; k < n->n_out; ++k)
            out->c[r * out->w + k] = table.c[r * table.w + n->out_idx[k]];
    rows_take_blobs(out, &table);
    rows_free(&table);
    return 0;
}

static size_t key_bytes_of(int32_t t) {
    switch (t) {
    case RJ_INT32: return 4;
    case RJ_INT64: return 8;
    case RJ_FP64: return 8;
    default: return 32; /* sizeof(std::string) in libstdc++ */
    }
}

static uint8_t tag_of_type(int32_t t) {
    switch (t) {
    case RJ_INT32: return T_I32;
    case RJ_INT64: return T_I64;
    case RJ_FP64: return T_F64;
    default: return T_STR;
    }
}

/* execute_hash_join + hash_join_omp<K> — reference src/execute.cpp:43-282 */
static int exec_join(const rj_plan* plan, const rj_node* n, rows* out, errbuf* e) {
    if (n->left >= plan->n_nodes || n->right >= plan->n_nodes) return fail(e, "join: bad child");
    /* key type = DataType of the BUILD side's key attribute (:271-273) */
    const rj_node* bn = &plan->nodes[n->build_left ? n->left : n->right];
    uint64_t       battr = n->build_left ? n->left_attr : n->right_attr;
    if (battr >= bn->n_out) return fail(e, "join: build attr out of range");
    int32_t kt = bn->out_type[battr];
    if (kt < RJ_INT32 || kt > RJ_VARCHAR) return fail(e, "Unsupported join type");
    uint8_t ktag = tag_of_type(kt);

    rows left, right;
    memset(&left, 0, sizeof left);
    memset(&right, 0, sizeof right);
    int rc = -1;
    /* 1) materialise both sides, left first (:48-49) */
    if (exec_node(plan, n->left, &left, e)) goto done;
    if (exec_node(plan, n->right, &right, e)) goto done;
    if (left.n == 0 || right.n == 0) { /* :50 */
        if (rows_init(out, 0, (size_t)n->n_out)) {
            fail(e, "oom");
            goto done;
        }
        rc = 0;
        goto done;
    }
    {
        const int build_left = n->build_left != 0;
        rows*     build = build_left ? &left : &right;
        rows*     probe = build_left ? &right : &left;
        size_t    bcol = (size_t)(build_left ? n->left_attr : n->right_attr);
        size_t    pcol = (size_t)(build_left ? n->right_attr : n->left_attr);
        size_t    left_w = left.w; /* left[0].size() (:57) */
        size_t    B = build->n, P = probe->n;
        if (bcol >= build->w || pcol >= probe->w) {
            fail(e, "join: key attr out of range");
            goto done;
        }
        for (uint64_t k = 0; k < n->n_out; ++k)
            if (n->out_idx[k] >= left.w + right.w) {
                fail(e, "join: output attr out of range");
                goto done;
            }

        /* 2) keys + valid: valid iff the variant holds exactly KeyType (:62-83) */
        uint8_t*  bvalid = (uint8_t*)calloc(B ? B : 1, 1);
        uint8_t*  pvalid = (uint8_t*)calloc(P ? P : 1, 1);
        uint64_t* bhash = (uint64_t*)malloc((B ? B : 1) * 8);
        uint64_t* phash = (uint64_t*)malloc((P ? P : 1) * 8);
        /* 3) bucket count (:86-92) */
        size_t    nb = (size_t)rjo_num_buckets(B, key_bytes_of(kt));
        size_t    bmask = nb - 1;
        uint32_t* bhist = (uint32_t*)calloc(nb + 1, 4);
        uint32_t* phist = (uint32_t*)calloc(nb + 1, 4);
        uint32_t* boff = (uint32_t*)calloc(nb + 1, 4);
        uint32_t* poff = (uint32_t*)calloc(nb + 1, 4);
        uint32_t* bbuf = (uint32_t*)malloc((B ? B : 1) * 4);
        uint32_t* pbuf = (uint32_t*)malloc((P ? P : 1) * 4);
        uint32_t* bo = (uint32_t*)malloc((nb + 1) * 4);
        uint32_t* po = (uint32_t*)malloc((nb + 1) * 4);
        /* output rows grow geometrically, like the per-thread vectors (:188) */
        size_t out_cap = 1024, out_n = 0, ow = (size_t)n->n_out;
        cell*  oc = (cell*)malloc(out_cap * (ow ? ow : 1) * sizeof(cell));
        /* per-bucket table storage, sized for the largest bucket below */
        cell*     slot_key = NULL;
        uint8_t*  slot_used = NULL;
        uint32_t *slot_head = NULL, *slot_tail = NULL, *chain = NULL;
        if (!bvalid || !pvalid || !bhash || !phash || !bhist || !phist || !boff || !poff || !bbuf ||
            !pbuf || !bo || !po || !oc) {
            fail(e, "oom");
            goto jdone;
        }
        for (size_t i = 0; i < B; ++i) {
            const cell* c = &build->c[i * build->w + bcol];
            if (c->tag == ktag) {
                bvalid[i] = 1;
                bhash[i] = hash_cell(c);
            }
        }
        for (size_t i = 0; i < P; ++i) {
            const cell* c = &probe->c[i * probe->w + pcol];
            if (c->tag == ktag) {
                pvalid[i] = 1;
                phash[i] = hash_cell(c);
            }
        }
        /* 4) histogram (:124-132), exclusive prefix (:169-174), scatter of row
         *    indices (:175-184) */
        for (size_t i = 0; i < B; ++i)
            if (bvalid[i]) bhist[bhash[i] & bmask]++;
        for (size_t i = 0; i < P; ++i)
            if (pvalid[i]) phist[phash[i] & bmask]++;
        for (size_t b = 0; b < nb; ++b) {
            boff[b + 1] = boff[b] + bhist[b];
            poff[b + 1] = poff[b] + phist[b];
        }
        memcpy(bo, boff, (nb + 1) * 4);
        memcpy(po, poff, (nb + 1) * 4);
        for (size_t i = 0; i < B; ++i)
            if (bvalid[i]) bbuf[bo[bhash[i] & bmask]++] = (uint32_t)i;
        for (size_t i = 0; i < P; ++i)
            if (pvalid[i]) pbuf[po[phash[i] & bmask]++] = (uint32_t)i;

        size_t max_cnt = 0;
        for (size_t b = 0; b < nb; ++b)
            if (bhist[b] > max_cnt) max_cnt = bhist[b];
        size_t max_cap = 1;
        while (max_cap < max_cnt * 2) max_cap <<= 1;
        slot_key = (cell*)malloc(max_cap * sizeof(cell));
        slot_used = (uint8_t*)malloc(max_cap);
        slot_head = (uint32_t*)malloc(max_cap * 4);
        slot_tail = (uint32_t*)malloc(max_cap * 4);
        chain = (uint32_t*)malloc((B ? B : 1) * 4); /* next build row with the same slot */
        if (!slot_key || !slot_used || !slot_head || !slot_tail || !chain) {
            fail(e, "oom");
            goto jdone;
        }
        const uint32_t NIL = 0xffffffffu;

        /* 5) per-bucket build + probe (:196-249) */
        for (size_t b = 0; b < nb; ++b) {
            uint32_t bs = boff[b], be = boff[b + 1];
            uint32_t ps = poff[b], pe = poff[b + 1];
            size_t   cnt = be - bs;
            if (cnt == 0 || ps == pe) continue; /* :200 */
            size_t cap = 1;
            while (cap < cnt * 2) cap <<= 1; /* :203-204 */
            size_t mask = cap - 1;
            memset(slot_used, 0, cap);
            /* build: linear probe; rows with equal keys are chained in arrival
             * order, as slot_idxs[h].push_back(row) does (:211-223) */
            for (uint32_t idx = bs; idx < be; ++idx) {
                uint32_t    row = bbuf[idx];
                const cell* key = &build->c[(size_t)row * build->w + bcol];
                size_t      h = (size_t)(bhash[row] & mask);
                while (slot_used[h] && !key_equal(&slot_key[h], key)) h = (h + 1) & mask;
                chain[row] = NIL;
                if (!slot_used[h]) {
                    slot_used[h] = 1;
                    slot_key[h] = *key;
                    slot_head[h] = row;
                } else {
                    chain[slot_tail[h]] = row;
                }
                slot_tail[h] = row;
            }
            /* probe (:226-248) */
            for (uint32_t idx = ps; idx < pe; ++idx) {
                uint32_t    prow = pbuf[idx];
                const cell* pkey = &probe->c[(size_t)prow * probe->w + pcol];
                size_t      h = (size_t)(phash[prow] & mask);
                while (slot_used[h]) {
                    if (key_equal(&slot_key[h], pkey)) {
                        for (uint32_t bi = slot_head[h]; bi != NIL; bi = chain[bi]) {
                            size_t L = build_left ? bi : prow; /* :233-234 */
                            size_t R = build_left ? prow : bi;
                            if (out_n == out_cap) {
                                out_cap *= 2;
                                cell* nc2 =
                                    (cell*)realloc(oc, out_cap * (ow ? ow : 1) * sizeof(cell));
                                if (!nc2) {
                                    fail(e, "oom");
                                    goto jdone;
                                }
                                oc = nc2;
                            }
                            for (size_t k = 0; k < ow; ++k) { /* :238-241 */
                                size_t ci = (size_t)n->out_idx[k];
                                oc[out_n * ow + k] = ci < left_w
                                                         ? left.c[L * left.w + ci]
                                                         : right.c[R * right.w + (ci - left_w)];
                            }
                            ++out_n;
                        }
                        break; /* :244 */
                    }
                    h = (h + 1) & mask;
                }
            }
        }
        /* 6) merge (:252-261): single thread here, so the concatenation is the
         * bucket order */
        memset(out, 0, sizeof *out);
        out->n = out_n;
        out->w = ow;
        out->c = oc;
        oc = NULL;
        rows_take_blobs(out, &left);
        rows_take_blobs(out, &right);
        rc = 0;
    jdone:
        free(bvalid); free(pvalid); free(bhash); free(phash); free(bhist); free(phist);
        free(boff); free(poff); free(bbuf); free(pbuf); free(bo); free(po); free(oc);
        free(slot_key); free(slot_used); free(slot_head); free(slot_tail); free(chain);
    }
done:
    rows_free(&left);
    rows_free(&right);
    return rc;
}

/* execute_impl — reference src/execute.cpp:302-314 */
static int exec_node(const rj_plan* plan, uint64_t node_idx, rows* out, errbuf* e) {
    if (node_idx >= plan->n_nodes) return fail(e, "bad node index");
    const rj_node* n = &plan->nodes[node_idx];
    if (n->kind == RJ_NODE_JOIN) return exec_join(plan, n, out, e);
    if (n->kind == RJ_NODE_SCAN) return exec_scan(plan, n, out, e);
    return fail(e, "bad node kind");
}

/* ----------------------------------------------------------- to_columnar --- */
typedef struct ocol {
    int32_t  type;
    uint8_t** pages;
    size_t   n, cap;
} ocol;

struct rjo_result {
    uint64_t num_rows;
    uint64_t n_cols;
    ocol*    cols;
};

static uint8_t* ocol_new_page(ocol* c) { /* Column::new_page, plan.h:64-68 */
    if (c->n == c->cap) {
        size_t    nc = c->cap ? c->cap * 2 : 4;
        uint8_t** np = (uint8_t**)realloc(c->pages, nc * sizeof(uint8_t*));
        if (!np) return NULL;
        c->pages = np;
        c->cap = nc;
    }
    uint8_t* p = (uint8_t*)malloc(RJ_PAGE_SIZE) /* malloc is 16-byte aligned; Page is alignas(8) */;
    if (!p) return NULL;
    memset(p, 0, RJ_PAGE_SIZE);
    c->pages[c->n++] = p;
    return p;
}

/* page under construction: values / offsets / chars / bitmap accumulate in
 * side buffers and are laid out by save_page (build_table.cpp:472-481,
 * :515-524, :558-567, :620-631) */
typedef struct pagebuf {
    uint16_t num_rows;
    size_t   n_vals;            /* fixed-width: values; varchar: offsets */
    uint8_t  vals[RJ_PAGE_SIZE];
    size_t   n_chars;
    char     chars[RJ_PAGE_SIZE];
    uint16_t offs[RJ_PAGE_SIZE / 2];
    uint8_t  bitmap[RJ_PAGE_SIZE];
    size_t   bitmap_bytes;
} pagebuf;

static void pb_reset(pagebuf* b) {
    b->num_rows = 0;
    b->n_vals = 0;
    b->n_chars = 0;
    b->bitmap_bytes = 0;
}

/* set_bitmap / unset_bitmap — build_table.cpp:438-454: the bitmap vector grows
 * to idx/8+1 bytes, zero-filled */
static void pb_bit(pagebuf* b, uint16_t idx, int set) {
    while (b->bitmap_bytes < (size_t)idx / 8 + 1) b->bitmap[b->bitmap_bytes++] = 0;
    if (set)
        b->bitmap[idx / 8] |= (uint8_t)(1u << (idx % 8));
    else
        b->bitmap[idx / 8] &= (uint8_t) ~(1u << (idx % 8));
}

static int pb_save_fixed(pagebuf* b, ocol* c, size_t vsz) {
    uint8_t* page = ocol_new_page(c);
    if (!page) return -1;
    uint16_t nv = (uint16_t)b->n_vals;
    memcpy(page, &b->num_rows, 2);
    memcpy(page + 2, &nv, 2);
    memcpy(page + (vsz == 4 ? 4 : 8), b->vals, b->n_vals * vsz);
    memcpy(page + RJ_PAGE_SIZE - b->bitmap_bytes, b->bitmap, b->bitmap_bytes);
    pb_reset(b);
    return 0;
}

static int pb_save_varchar(pagebuf* b, ocol* c) {
    uint8_t* page = ocol_new_page(c);
    if (!page) return -1;
    uint16_t nv = (uint16_t)b->n_vals;
    memcpy(page, &b->num_rows, 2);
    memcpy(page + 2, &nv, 2);
    memcpy(page + 4, b->offs, b->n_vals * 2);
    memcpy(page + 4 + b->n_vals * 2, b->chars, b->n_chars);
    memcpy(page + RJ_PAGE_SIZE - b->bitmap_bytes, b->bitmap, b->bitmap_bytes);
    pb_reset(b);
    return 0;
}

/* save_long_string — build_table.cpp:603-619 */
static int save_long_string(ocol* c, const char* s, size_t len) {
    size_t off = 0;
    int    first = 1;
    while (off < len) {
        uint8_t* page = ocol_new_page(c);
        if (!page) return -1;
        uint16_t tag = first ? 0xffff : 0xfffe;
        first = 0;
        size_t   chunk = len - off < RJ_PAGE_SIZE - 4 ? len - off : RJ_PAGE_SIZE - 4;
        uint16_t n16 = (uint16_t)chunk;
        memcpy(page, &tag, 2);
        memcpy(page + 2, &n16, 2);
        memcpy(page + 4, s + off, chunk);
        off += chunk;
    }
    return 0;
}

/* One column of Table::to_columnar — reference build_table.cpp:462-678.
 * `get(i)` is column `ci` of row i.                                          */
static int encode_column(const rows* t, size_t ci, int32_t type, ocol* c, errbuf* e) {
    pagebuf* b = (pagebuf*)malloc(sizeof(pagebuf));
    if (!b) return fail(e, "oom");
    pb_reset(b);
    c->type = type;
    int rc = 0;
    for (size_t r = 0; r < t->n && rc == 0; ++r) {
        const cell* v = &t->c[r * t->w + ci];
        switch (type) {
        case RJ_INT32:
        case RJ_INT64:
        case RJ_FP64: {
            size_t  vsz = type == RJ_INT32 ? 4 : 8;
            size_t  hdr = type == RJ_INT32 ? 4 : 8;
            uint8_t want = tag_of_type(type);
            if (v->tag == want) {
                /* :488 / :531 / :574 */
                if (hdr + (b->n_vals + 1) * vsz + (b->num_rows / 8 + 1) > RJ_PAGE_SIZE)
                    if (pb_save_fixed(b, c, vsz)) rc = fail(e, "oom");
                pb_bit(b, b->num_rows, 1);
                if (type == RJ_INT32)
                    memcpy(b->vals + b->n_vals * 4, &v->v.i32, 4);
                else if (type == RJ_INT64)
                    memcpy(b->vals + b->n_vals * 8, &v->v.i64, 8);
                else
                    memcpy(b->vals + b->n_vals * 8, &v->v.f64, 8);
                b->n_vals++;
                b->num_rows++;
            } else if (v->tag == T_NULL) {
                /* :495 / :538 / :581 */
                if (hdr + b->n_vals * vsz + (b->num_rows / 8 + 1) > RJ_PAGE_SIZE)
                    if (pb_save_fixed(b, c, vsz)) rc = fail(e, "oom");
                pb_bit(b, b->num_rows, 0);
                b->num_rows++;
            }
            /* any other alternative is silently skipped by the visitor */
            break;
        }
        case RJ_VARCHAR: {
            if (v->tag == T_STR) {
                if (v->len > RJ_PAGE_SIZE - 7) { /* :644-648 */
                    if (b->num_rows > 0 && pb_save_varchar(b, c)) rc = fail(e, "oom");
                    if (rc == 0 && save_long_string(c, v->v.s, v->len)) rc = fail(e, "oom");
                } else {
                    /* :650-653 */
                    if (4 + (b->n_vals + 1) * 2 + (b->n_chars + v->len) + (b->num_rows / 8 + 1) >
                        RJ_PAGE_SIZE)
                        if (pb_save_varchar(b, c)) rc = fail(e, "oom");
                    pb_bit(b, b->num_rows, 1);
                    memcpy(b->chars + b->n_chars, v->v.s, v->len);
                    b->n_chars += v->len;
                    b->offs[b->n_vals++] = (uint16_t)b->n_chars;
                    b->num_rows++;
                }
            } else if (v->tag == T_NULL) {
                /* :661-663 */
                if (4 + b->n_vals * 2 + b->n_chars + (b->num_rows / 8 + 1) > RJ_PAGE_SIZE)
                    if (pb_save_varchar(b, c)) rc = fail(e, "oom");
                pb_bit(b, b->num_rows, 0);
                b->num_rows++;
            } else {
                rc = fail(e, "not string or null"); /* :668 */
            }
            break;
        }
        default: rc = fail(e, "unknown column type %d", type);
        }
    }
    if (rc == 0 && b->num_rows != 0) {
        if (type == RJ_VARCHAR) {
            if (pb_save_varchar(b, c)) rc = fail(e, "oom");
        } else if (pb_save_fixed(b, c, type == RJ_INT32 ? 4 : 8))
            rc = fail(e, "oom");
    }
    free(b);
    return rc;
}

void rjo_result_free(rjo_result* r) {
    if (!r) return;
    for (uint64_t c = 0; c < r->n_cols; ++c) {
        for (size_t p = 0; p < r->cols[c].n; ++p) free(r->cols[c].pages[p]);
        free(r->cols[c].pages);
    }
    free(r->cols);
    free(r);
}

static rjo_result* result_new(uint64_t n_cols) {
    rjo_result* r = (rjo_result*)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->n_cols = n_cols;
    r->cols = (ocol*)calloc(n_cols ? n_cols : 1, sizeof(ocol));
    if (!r->cols) {
        free(r);
        return NULL;
    }
    return r;
}

/* execute — reference src/execute.cpp:316-324 */
int rjo_execute(const rj_plan* plan, rjo_result** out, char* err, size_t errcap) {
    errbuf e = {err, errcap};
    if (err && errcap) err[0] = 0;
    if (!plan || !out) return fail(&e, "null argument");
    if (plan->root >= plan->n_nodes) return fail(&e, "bad root");
    rows ret;
    memset(&ret, 0, sizeof ret);
    if (exec_node(plan, plan->root, &ret, &e)) {
        rows_free(&ret);
        return -1;
    }
    const rj_node* root = &plan->nodes[plan->root];
    rjo_result*    r = result_new(root->n_out);
    if (!r) {
        rows_free(&ret);
        return fail(&e, "oom");
    }
    r->num_rows = ret.n; /* ret.num_rows = table.size(), build_table.cpp:461 */
    for (uint64_t c = 0; c < root->n_out; ++c) {
        if (encode_column(&ret, (size_t)c, root->out_type[c], &r->cols[c], &e)) {
            rjo_result_free(r);
            rows_free(&ret);
            return -1;
        }
    }
    rows_free(&ret);
    *out = r;
    return 0;
}

uint64_t rjo_result_num_rows(const rjo_result* r) { return r->num_rows; }
uint64_t rjo_result_num_cols(const rjo_result* r) { return r->n_cols; }
int32_t  rjo_result_col_type(const rjo_result* r, uint64_t c) { return r->cols[c].type; }
uint64_t rjo_result_col_pages(const rjo_result* r, uint64_t c) { return r->cols[c].n; }
const void* rjo_result_page(const rjo_result* r, uint64_t c, uint64_t p) {
    return r->cols[c].pages[p];
}
void rjo_free(void* p) { free(p); }

/* ------------------------------------------------- single-column helpers --- */
int rjo_decode_fixed(const rj_column* col, uint64_t num_rows, void* values, uint8_t* valid,
                     char* err, size_t errcap) {
    errbuf e = {err, errcap};
    if (err && errcap) err[0] = 0;
    if (col->type == RJ_VARCHAR) return fail(&e, "rjo_decode_fixed on VARCHAR");
    rows t;
    if (rows_init(&t, (size_t)num_rows, 1)) return fail(&e, "oom");
    if (decode_column(col, num_rows, &t, 0, &e)) {
        rows_free(&t);
        return -1;
    }
    size_t vsz = col->type == RJ_INT32 ? 4 : 8;
    for (size_t i = 0; i < t.n; ++i) {
        valid[i] = t.c[i].tag != T_NULL;
        if (col->type == RJ_INT32)
            memcpy((char*)values + i * vsz, &t.c[i].v.i32, 4);
        else
            memcpy((char*)values + i * vsz, &t.c[i].v.i64, 8);
        if (!valid[i]) memset((char*)values + i * vsz, 0, vsz);
    }
    rows_free(&t);
    return 0;
}

int rjo_decode_varchar(const rj_column* col, uint64_t num_rows, uint64_t* offsets, uint8_t* valid,
                       char** heap, char* err, size_t errcap) {
    errbuf e = {err, errcap};
    if (err && errcap) err[0] = 0;
    rows t;
    if (rows_init(&t, (size_t)num_rows, 1)) return fail(&e, "oom");
    if (decode_column(col, num_rows, &t, 0, &e)) {
        rows_free(&t);
        return -1;
    }
    uint64_t total = 0;
    for (size_t i = 0; i < t.n; ++i)
        if (t.c[i].tag == T_STR) total += t.c[i].len;
    char* h = (char*)malloc(total ? total : 1);
    if (!h) {
        rows_free(&t);
        return fail(&e, "oom");
    }
    uint64_t off = 0;
    for (size_t i = 0; i < t.n; ++i) {
        offsets[i] = off;
        valid[i] = t.c[i].tag == T_STR;
        if (valid[i]) {
            memcpy(h + off, t.c[i].v.s, t.c[i].len);
            off += t.c[i].len;
        }
    }
    offsets[t.n] = off;
    *heap = h;
    rows_free(&t);
    return 0;
}

int rjo_encode_fixed(int32_t type, const void* values, const uint8_t* valid, uint64_t n,
                     rjo_result** out) {
    if (type != RJ_INT32 && type != RJ_INT64 && type != RJ_FP64) return -1;
    rows t;
    if (rows_init(&t, (size_t)n, 1)) return -1;
    size_t vsz = type == RJ_INT32 ? 4 : 8;
    for (size_t i = 0; i < t.n; ++i) {
        if (valid && !valid[i]) continue;
        t.c[i].tag = tag_of_type(type);
        if (type == RJ_INT32)
            memcpy(&t.c[i].v.i32, (const char*)values + i * vsz, 4);
        else
            memcpy(&t.c[i].v.i64, (const char*)values + i * vsz, 8);
    }
    rjo_result* r = result_new(1);
    if (!r) {
        rows_free(&t);
        return -1;
    }
    r->num_rows = n;
    int rc = encode_column(&t, 0, type, &r->cols[0], NULL);
    rows_free(&t);
    if (rc) {
        rjo_result_free(r);
        return -1;
    }
    *out = r;
    return 0;
}

int rjo_encode_varchar(const uint64_t* offsets, const char* heap, const uint8_t* valid, uint64_t n,
                       rjo_result** out) {
    rows t;
    if (rows_init(&t, (size_t)n, 1)) return -1;
    for (size_t i = 0; i < t.n; ++i) {
        if (valid && !valid[i]) continue;
        t.c[i].tag = T_STR;
        t.c[i].len = (uint32_t)(offsets[i + 1] - offsets[i]);
        t.c[i].v.s = heap + offsets[i];
    }
    rjo_result* r = result_new(1);
    if (!r) {
        rows_free(&t);
        return -1;
    }
    r->num_rows = n;
    int rc = encode_column(&t, 0, RJ_VARCHAR, &r->cols[0], NULL);
    rows_free(&t);
    if (rc) {
        rjo_result_free(r);
        return -1;
    }
    *out = r;
    return 0;
}
