// vf_arrow.cpp -- Arrow C Data Interface export of the result frame (zero-copy hand-off to polars /
// pyarrow; replaces the Vec<String>/Vec<u64> -> df! step of src/lib.rs:312-319).
// Exports a struct array {sequence: large_utf8, count: uint64}; the exported arrays own the buffers.
#include "../../include/vfind_hip.h"
#include "vf_internal.h"

#include <cstdlib>
#include <cstring>

namespace {

struct Holder {                 // shared by parent and children; freed when the last one is released
    vf_result data;
    int refs = 0;
};
struct Priv {
    Holder *holder;
    const void *buffers[3];
    ArrowArray *children[2];
    ArrowArray child_storage[2];
};

void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; i < s->n_children; ++i) {
        if (s->children[i]->release) s->children[i]->release(s->children[i]);
        free(s->children[i]);
    }
    free(s->children);
    s->release = nullptr;
}
void release_child_schema(ArrowSchema *s) { if (s) s->release = nullptr; }

void drop(Holder *h) { if (--h->refs == 0) delete h; }
void release_child(ArrowArray *a) {
    if (!a || !a->release) return;
    Priv *p = (Priv *)a->private_data;
    drop(p->holder);
    delete p;
    a->release = nullptr;
}
void release_parent(ArrowArray *a) {
    if (!a || !a->release) return;
    Priv *p = (Priv *)a->private_data;
    for (int i = 0; i < 2; ++i)
        if (p->children[i]->release) p->children[i]->release(p->children[i]);
    drop(p->holder);
    delete p;
    a->release = nullptr;
}

ArrowSchema *child_schema(const char *fmt, const char *name) {
    ArrowSchema *c = (ArrowSchema *)calloc(1, sizeof(ArrowSchema));
    c->format = fmt; c->name = name; c->flags = 0;       // not nullable
    c->release = release_child_schema;
    return c;
}

}  // namespace

static int vf_result_export_arrow_impl(vf_result *r, ArrowArray *out_array, ArrowSchema *out_schema);
extern "C" int vf_result_export_arrow(vf_result *r, ArrowArray *out_array, ArrowSchema *out_schema) { return vf_guard([&] { return vf_result_export_arrow_impl(r, out_array, out_schema); }); }
static int vf_result_export_arrow_impl(vf_result *r, ArrowArray *out_array, ArrowSchema *out_schema) {
    if (!r || !out_array || !out_schema) return vf_internal_fail(VF_EINVAL, "null argument");
    const int64_t rows = (int64_t)r->counts.size();
    if (r->offsets.size() != (size_t)rows + 1) return vf_internal_fail(VF_EINVAL, "result already exported");

    memset(out_schema, 0, sizeof(*out_schema));
    out_schema->format = "+s";
    out_schema->name = "";
    out_schema->n_children = 2;
    out_schema->children = (ArrowSchema **)calloc(2, sizeof(ArrowSchema *));
    out_schema->children[0] = child_schema("U", "sequence");     // large_utf8, int64 offsets
    out_schema->children[1] = child_schema("L", "count");        // uint64
    out_schema->release = release_schema;

    Holder *h = new Holder;
    h->data.keys.swap(r->keys);
    h->data.offsets.swap(r->offsets);
    h->data.counts.swap(r->counts);
    if (h->data.keys.empty()) h->data.keys.push_back(0);          // non-null data pointers
    if (h->data.counts.empty()) h->data.counts.push_back(0);
    h->refs = 3;

    Priv *pp = new Priv;
    pp->holder = h;
    Priv *pc[2] = {new Priv, new Priv};
    for (int i = 0; i < 2; ++i) { pc[i]->holder = h; }
    // sequence: validity (none), offsets, data
    pc[0]->buffers[0] = nullptr; pc[0]->buffers[1] = h->data.offsets.data(); pc[0]->buffers[2] = h->data.keys.data();
    pc[1]->buffers[0] = nullptr; pc[1]->buffers[1] = h->data.counts.data(); pc[1]->buffers[2] = nullptr;
    for (int i = 0; i < 2; ++i) {
        ArrowArray *c = &pp->child_storage[i];
        memset(c, 0, sizeof(*c));
        c->length = rows; c->null_count = 0; c->offset = 0;
        c->n_buffers = i == 0 ? 3 : 2;
        c->buffers = pc[i]->buffers;
        c->release = release_child;
        c->private_data = pc[i];
        pp->children[i] = c;
    }
    memset(out_array, 0, sizeof(*out_array));
    out_array->length = rows; out_array->null_count = 0; out_array->offset = 0;
    out_array->n_buffers = 1;
    pp->buffers[0] = nullptr;
    out_array->buffers = pp->buffers;
    out_array->n_children = 2;
    out_array->children = pp->children;
    out_array->release = release_parent;
    out_array->private_data = pp;
    return VF_OK;
}
