/*
 * tpch_rows.c -- drives the reference's vendored TPC-H generator (third_party/tpch-dbgen, plain C) the way Hyrise's
 * TPCHTableGenerator::generate() does (src/benchmarklib/tpch/tpch_table_generator.cpp:141-316): dbgen_reset_seeds(),
 * dbgen_init_scale_factor(sf), then per table  row_start(t); mk_*(i, &row); row_stop(t)  (call_dbgen_mk, :71-88) -- customers first,
 * then orders with their lineitems -- and writes the columns the hot path reads as little-endian binary arrays.
 *
 * TEST INFRASTRUCTURE: compiled by oracle/Makefile from the sources where they lie under /root/reference (never copied) into
 * oracle/_ref/tpch_rows; only tests/ and tools/make_dbgen_fixture.py run it.  Nothing in the product path does.
 *
 * usage: tpch_rows <scale factor> <output file>
 * output: "HYDBGEN1" | u64 orders | u64 lineitems | i32 o_orderkey[orders] | i32 l_orderkey[n] | i32 l_quantity[n] | i64 l_extendedprice
 *         cents[n] | i32 l_discount cents[n] | i32 l_tax cents[n] | u8 l_returnflag[n] | u8 l_linestatus[n] | char l_shipdate[n][10]
 *         | char l_commitdate[n][10] | char l_receiptdate[n][10]          (n = lineitems; dates as dbgen prints them: YYYY-MM-DD)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "config.h"
#include "dss.h"
#include "dsstypes.h"
#include "tpch_dbgen.h"





typedef struct {
  size_t n, capacity, width;
  char* data;
} column;

static void push(column* c, const void* value) {
  if (c->n == c->capacity) {
    c->capacity = c->capacity ? c->capacity * 2 : 1u << 16;
    c->data = (char*)realloc(c->data, c->capacity * c->width);
    if (!c->data) { fprintf(stderr, "out of memory\n"); exit(1); }
  }
  memcpy(c->data + c->n * c->width, value, c->width);
  c->n += 1;
}

static void put(FILE* out, const column* c) {
  if (c->n && fwrite(c->data, c->width, c->n, out) != c->n) { fprintf(stderr, "write failed\n"); exit(1); }
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <scale factor> <output file>\n", argv[0]); return 2; }
  const float scale_factor = (float)atof(argv[1]);
  dbgen_reset_seeds();
  dbgen_init_scale_factor(scale_factor);
  const size_t customer_count = (size_t)(tdefs[CUST].base * scale);
  const size_t order_count = (size_t)(tdefs[ORDER].base * scale);
  /* Hyrise generates the customers before the orders (:176-186): the tables' random streams are separate, the calls are kept anyway */
  for (size_t i = 0; i < customer_count; ++i) {
    customer_t customer;
    row_start(CUST);
    mk_cust((DSS_HUGE)(i + 1), &customer);
    row_stop(CUST);
  }
  column o_orderkey = {0, 0, 4, NULL}, l_orderkey = {0, 0, 4, NULL}, l_quantity = {0, 0, 4, NULL}, l_extendedprice = {0, 0, 8, NULL};
  column l_discount = {0, 0, 4, NULL}, l_tax = {0, 0, 4, NULL}, l_returnflag = {0, 0, 1, NULL}, l_linestatus = {0, 0, 1, NULL};
  column l_shipdate = {0, 0, 10, NULL}, l_commitdate = {0, 0, 10, NULL}, l_receiptdate = {0, 0, 10, NULL};
  for (size_t i = 0; i < order_count; ++i) {
    order_t order;
    memset(&order, 0, sizeof(order));
    row_start(ORDER);
    mk_order((DSS_HUGE)(i + 1), &order, 0L);
    row_stop(ORDER);
    const int32_t okey = (int32_t)order.okey;
    push(&o_orderkey, &okey);
    for (DSS_HUGE l = 0; l < order.lines; ++l) {
      const line_t* line = &order.l[l];
      const int32_t key = (int32_t)line->okey, quantity = (int32_t)line->quantity, discount = (int32_t)line->discount, tax = (int32_t)line->tax;
      const int64_t price = (int64_t)line->eprice;
      push(&l_orderkey, &key);
      push(&l_quantity, &quantity);
      push(&l_extendedprice, &price);
      push(&l_discount, &discount);
      push(&l_tax, &tax);
      push(&l_returnflag, &line->rflag[0]);
      push(&l_linestatus, &line->lstatus[0]);
      push(&l_shipdate, line->sdate);
      push(&l_commitdate, line->cdate);
      push(&l_receiptdate, line->rdate);
    }
  }
  FILE* out = fopen(argv[2], "wb");
  if (!out) { perror(argv[2]); return 1; }
  const uint64_t counts[2] = {o_orderkey.n, l_orderkey.n};
  fwrite("HYDBGEN1", 1, 8, out);
  fwrite(counts, 8, 2, out);
  put(out, &o_orderkey);
  put(out, &l_orderkey);
  put(out, &l_quantity);
  put(out, &l_extendedprice);
  put(out, &l_discount);
  put(out, &l_tax);
  put(out, &l_returnflag);
  put(out, &l_linestatus);
  put(out, &l_shipdate);
  put(out, &l_commitdate);
  put(out, &l_receiptdate);
  fclose(out);
  fprintf(stderr, "%zu orders, %zu lineitems\n", o_orderkey.n, l_orderkey.n);
  return 0;
}
