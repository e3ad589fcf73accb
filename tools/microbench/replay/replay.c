/* Experiment helper (not product code): replays a prepared list of library launches from C, so that the host costs ~1 us per
 * launch instead of the ~17 us of a Python call — to see what the GPU does with PlanningEnv's inner iterations of several row
 * groups enqueued on several plain streams.   gcc -O2 -shared -fPIC -o replay.so replay.c */
typedef int (*step_fn)(void *, long long, const void *, void *);
typedef int (*actor_fn)(const float *, long long, long long, const float *, const float *, const float *, float *, float *, int, void *);
struct item {
    int kind, device;              /* 0: np_f16_step, 1: np_actor_forward */
    void *ctx; long long n; const void *io; void *stream;
    const float *w; long long nf; const float *obs, *hin, *mask; float *act, *hout;
};
int replay(step_fn step, actor_fn actor, const struct item *it, int count) {
    for (int i = 0; i < count; i++) {
        int rc = it[i].kind == 0 ? step(it[i].ctx, it[i].n, it[i].io, it[i].stream)
                                 : actor(it[i].w, it[i].nf, it[i].n, it[i].obs, it[i].hin, it[i].mask, it[i].act, it[i].hout, it[i].device, it[i].stream);
        if (rc) return 1000 + i;
    }
    return 0;
}
