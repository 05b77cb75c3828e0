/*
 * chain.c -- oracle restatement of signal/signal.go (ordered slots of effects units with
 * bypass flags and ping-pong buffers).  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 * PARITY UNPINNED by the reference (signal/ has no tests upstream).
 */
#include "gdg_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { gdgo_unit *unit; int bypass; } slot_t;

struct gdgo_chain {
    slot_t *slots; int n_slots;
    double *buf_in, *buf_out; int buf_n;
};

/* signal/signal.go:419-431 */
gdgo_chain *gdgo_chain_create(void) { return (gdgo_chain *)calloc(1, sizeof(gdgo_chain)); }

void gdgo_chain_destroy(gdgo_chain *c) {
    if (!c) return;
    for (int i = 0; i < c->n_slots; i++) gdgo_unit_destroy(c->slots[i].unit);
    free(c->slots); free(c->buf_in); free(c->buf_out);
    free(c);
}

/* signal/signal.go:52-86: a new unit is appended in BYPASS mode */
int gdgo_chain_append_unit(gdgo_chain *c, int unit_type) {
    gdgo_unit *u = gdgo_unit_create(unit_type);
    if (!u) return -1;
    c->slots = (slot_t *)realloc(c->slots, sizeof(slot_t) * (size_t)(c->n_slots + 1));
    c->slots[c->n_slots].unit = u;
    c->slots[c->n_slots].bypass = 1;
    return c->n_slots++;
}

/* signal/signal.go:91-113 */
int gdgo_chain_remove_unit(gdgo_chain *c, int id) {
    if (id < 0 || id >= c->n_slots) return -1;
    gdgo_unit_destroy(c->slots[id].unit);
    memmove(c->slots + id, c->slots + id + 1, sizeof(slot_t) * (size_t)(c->n_slots - id - 1));
    c->n_slots--;
    return 0;
}

/* signal/signal.go:118-135: swap with the predecessor; state travels with the unit */
int gdgo_chain_move_up(gdgo_chain *c, int id) {
    if (id < 1 || id >= c->n_slots) return -1;
    slot_t t = c->slots[id]; c->slots[id] = c->slots[id - 1]; c->slots[id - 1] = t;
    return 0;
}

/* signal/signal.go:140-157 */
int gdgo_chain_move_down(gdgo_chain *c, int id) {
    if (id < 0 || id >= c->n_slots - 1) return -1;
    slot_t t = c->slots[id]; c->slots[id] = c->slots[id + 1]; c->slots[id + 1] = t;
    return 0;
}

int gdgo_chain_set_bypass(gdgo_chain *c, int id, int bypass) {
    if (id < 0 || id >= c->n_slots) return -1;
    c->slots[id].bypass = bypass ? 1 : 0;
    return 0;
}

int gdgo_chain_length(const gdgo_chain *c) { return c->n_slots; }

gdgo_unit *gdgo_chain_unit(gdgo_chain *c, int id) {
    if (id < 0 || id >= c->n_slots) return NULL;
    return c->slots[id].unit;
}

/* signal/signal.go:361-414: length mismatch is a silent no-op; bypassed slots do not advance their state */
void gdgo_chain_process(gdgo_chain *c, const double *in, int n_in, double *out, int n_out, uint32_t sample_rate) {
    if (n_in != n_out) return;
    int n = n_in;
    if (c->buf_n != n || c->buf_in == NULL) {
        free(c->buf_in); free(c->buf_out);
        c->buf_in = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        c->buf_out = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        c->buf_n = n;
    }
    double *a = c->buf_in, *b = c->buf_out;
    memcpy(a, in, sizeof(double) * (size_t)n);
    for (int i = 0; i < c->n_slots; i++) {
        if (!c->slots[i].bypass) {
            gdgo_unit_process(c->slots[i].unit, a, b, n, sample_rate);
            double *t = a; a = b; b = t;
        }
    }
    c->buf_in = a;
    c->buf_out = b;
    memcpy(out, a, sizeof(double) * (size_t)n);
}
