/*
 * rjo_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's hot path, row-at-a-time exactly as
 * the reference does it:
 *   Table::from_columnar  (reference src/build_table.cpp:312-436)
 *   execute_scan          (reference src/execute.cpp:284-300)
 *   execute_hash_join / hash_join_omp (reference src/execute.cpp:43-282)
 *   Table::to_columnar    (reference src/build_table.cpp:456-681)
 *   execute               (reference src/execute.cpp:316-324)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (librj.so) never links or calls it.
 *
 * Parity pin: the reference itself is NOT buildable in this image (its hot-path
 * TUs include range-v3 0.12.0 and RE2 headers, which are absent, and stand-ins
 * are not allowed), so this restatement is pinned by the reference's own
 * known-answer tests: the 8 cases of reference tests/unit_tests.cpp, committed
 * as data in tests/golden/unit_cases.json (tests/test_oracle_golden.py).
 */
#ifndef RJO_ORACLE_H_
#define RJO_ORACLE_H_

#include "../include/rj.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rjo_result rjo_result;

/* execute(): returns 0 on success; on failure writes a message into err. */
int rjo_execute(const rj_plan* plan, rjo_result** out, char* err, size_t errcap);

uint64_t    rjo_result_num_rows(const rjo_result* r);
uint64_t    rjo_result_num_cols(const rjo_result* r);
int32_t     rjo_result_col_type(const rjo_result* r, uint64_t col);
uint64_t    rjo_result_col_pages(const rjo_result* r, uint64_t col);
const void* rjo_result_page(const rjo_result* r, uint64_t col, uint64_t page);
void        rjo_result_free(rjo_result* r);

/* from_columnar on ONE fixed-width column: values[i] (4 or 8 bytes each by
 * type) and valid[i] for i < num_rows.  Returns 0, or -1 with err set
 * ("row_idx": pages hold more rows than num_rows, build_table.cpp:334).     */
int rjo_decode_fixed(const rj_column* col, uint64_t num_rows, void* values, uint8_t* valid,
                     char* err, size_t errcap);
/* from_columnar on ONE VARCHAR column: offsets[num_rows+1] into a byte heap
 * allocated by the oracle (*heap, free with rjo_free).                       */
int rjo_decode_varchar(const rj_column* col, uint64_t num_rows, uint64_t* offsets, uint8_t* valid,
                       char** heap, char* err, size_t errcap);
void rjo_free(void* p);

/* to_columnar on ONE column: a 1-column rjo_result whose pages follow the
 * reference's page-fill rules.  `values` is 4/8 bytes per row for fixed-width
 * types; for VARCHAR pass offsets[n+1] + heap.                               */
int rjo_encode_fixed(int32_t type, const void* values, const uint8_t* valid, uint64_t n,
                     rjo_result** out);
int rjo_encode_varchar(const uint64_t* offsets, const char* heap, const uint8_t* valid,
                       uint64_t n, rjo_result** out);

/* Table::from_csv (reference src/build_table.cpp:135-304): CSVParser::execute / finish
 * (src/csv_parser.cpp:3-175, escape '\\', separator ','), TableParser::on_field (:31-76: empty field
 * = NULL, std::from_chars for integers), the filter as bitmap arithmetic (src/statement.cpp:8-135,
 * 186-201; include/inner_column.h:170-324) and from_inner_to_column with ColumnInserter's
 * page-fill rule (:94-119, include/plan.h:151-335).  FP64 fields: std::from_chars(double) restated
 * (longest-prefix grammar, nearest double via strtod, out-of-range = error; FP64 comparison leaves
 * carry the literal's bits in ivalue).
 * Returns 0, or -1 with err set to the reference's message.                                     */
int rjo_from_csv(const char* text, uint64_t n_bytes, uint64_t n_cols, const int32_t* col_type,
                 const rj_filter_op* filter, uint64_t n_filter_ops, rjo_result** out, char* err, size_t errcap);

/* HashUtil<K>::hash for integral keys (src/execute.cpp:19-27) — exposed so the
 * tests can pin the restatement of the bucket rule.                          */
uint64_t rjo_hash_int(int64_t key);
/* bucket-count rule (src/execute.cpp:86-92) */
uint64_t rjo_num_buckets(uint64_t build_rows, uint64_t key_bytes);

#ifdef __cplusplus
}
#endif
#endif
