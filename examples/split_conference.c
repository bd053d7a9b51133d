/*
 * examples/split_conference.c -- one process, every GPU of the node, in C: conferences whose members are spread over the
 * GPUs, mixed every 10 ms through the path's one exchange step (src/audiofilters/audiomixer.c:304-314 across devices):
 *
 *   mi_mixer_partial_sum (local members, int32)  ->  mi_exchange_allreduce_i32 (RCCL over xGMI)  ->  mi_mixer_finalize
 *
 * One thread per GPU, as a mediastreamer2 process runs one ticker thread per conference (src/voip/audioconference.c:72);
 * each thread owns a context on its device, its share of every conference's members and one rank of the exchange.  Thread
 * 0 also mixes the whole conferences on its own GPU (mi_mixer_process) and checks that what the exchange delivered to its
 * members is the same, bit for bit -- integer addition does not care how the sum was split.
 *
 *   cc -std=c99 -Iinclude examples/split_conference.c -Lmediastreamer2_amd -lmsmi355x -lpthread -Wl,-rpath,$PWD/mediastreamer2_amd
 *   ./a.out [gpus]      (default: every visible device that divides 32; prints "ok <gpus>")
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msmi355x.h"

enum { CONFERENCES = 64, MEMBERS = 32, SAMPLES = 480, TICKS = 20 };

typedef struct {
	int rank, nranks, ok;
	unsigned char id[MI_EXCHANGE_ID_BYTES];
	char err[256];
} rank_t;

/* member m of conference c at tick t: any deterministic signal every rank can regenerate */
static int16_t sample_of(int c, int m, int t, int i) {
	uint32_t x = (uint32_t)(((c * MEMBERS + m) * 131 + t) * 4099 + i) * 2654435761u;
	return (int16_t)((int)(x >> 17) - 16384); /* +-16384: 32 of them saturate the mix now and then */
}

static void *rank_main(void *arg) {
	rank_t *r = (rank_t *)arg;
	const int mloc = MEMBERS / r->nranks, first = r->rank * mloc;
	const size_t nloc = (size_t)CONFERENCES * mloc * SAMPLES, nall = (size_t)CONFERENCES * MEMBERS * SAMPLES;
	mi_ctx *ctx = NULL;
	mi_mixer *mine = NULL, *whole = NULL;
	mi_exchange *x = NULL;
	int16_t *h_loc = malloc(nloc * 2), *h_out = malloc(nloc * 2), *h_all = NULL, *h_ref = NULL;
	int16_t *d_in = NULL, *d_out = NULL, *d_all = NULL, *d_ref = NULL;
	int32_t *d_sum = NULL;
	int t, c, m, i;
	r->ok = 0;
#define MUST(call)                                                                    \
	do {                                                                              \
		if ((call) != MI_OK) {                                                        \
			snprintf(r->err, sizeof r->err, "rank %d: %s: %s", r->rank, #call, mi_last_error()); \
			goto done;                                                                \
		}                                                                             \
	} while (0)
	MUST(mi_ctx_create(r->rank, NULL, &ctx));
	MUST(mi_mixer_create(ctx, CONFERENCES, mloc, SAMPLES, &mine));
	MUST(mi_exchange_create(ctx, r->nranks, r->rank, r->id, &x)); /* returns when every rank has joined */
	d_in = mi_dev_alloc(ctx, nloc * 2), d_out = mi_dev_alloc(ctx, nloc * 2);
	d_sum = mi_dev_alloc(ctx, (size_t)CONFERENCES * SAMPLES * 4);
	if (r->rank == 0) {
		MUST(mi_mixer_create(ctx, CONFERENCES, MEMBERS, SAMPLES, &whole));
		h_all = malloc(nall * 2), h_ref = malloc(nall * 2);
		d_all = mi_dev_alloc(ctx, nall * 2), d_ref = mi_dev_alloc(ctx, nall * 2);
	}
	for (t = 0; t < TICKS; ++t) {
		for (c = 0; c < CONFERENCES; ++c)
			for (m = 0; m < mloc; ++m)
				for (i = 0; i < SAMPLES; ++i) h_loc[((size_t)c * mloc + m) * SAMPLES + i] = sample_of(c, first + m, t, i);
		MUST(mi_copy_h2d(ctx, d_in, h_loc, nloc * 2));
		MUST(mi_mixer_partial_sum(mine, d_in, NULL, d_sum));
		MUST(mi_exchange_allreduce_i32(x, d_sum, (size_t)CONFERENCES * SAMPLES)); /* on the context's stream: ordered by it */
		MUST(mi_mixer_finalize(mine, d_in, NULL, d_sum, 1, d_out));
		MUST(mi_copy_d2h(ctx, h_out, d_out, nloc * 2));
		if (r->rank == 0) { /* the same conferences mixed whole on this GPU */
			for (c = 0; c < CONFERENCES; ++c)
				for (m = 0; m < MEMBERS; ++m)
					for (i = 0; i < SAMPLES; ++i) h_all[((size_t)c * MEMBERS + m) * SAMPLES + i] = sample_of(c, m, t, i);
			MUST(mi_copy_h2d(ctx, d_all, h_all, nall * 2));
			MUST(mi_mixer_process(whole, d_all, NULL, 1, d_ref));
			MUST(mi_copy_d2h(ctx, h_ref, d_ref, nall * 2));
		}
		MUST(mi_ctx_sync(ctx));
		if (r->rank == 0)
			for (c = 0; c < CONFERENCES; ++c)
				if (memcmp(h_out + (size_t)c * mloc * SAMPLES, h_ref + (size_t)c * MEMBERS * SAMPLES, (size_t)mloc * SAMPLES * 2) != 0) {
					snprintf(r->err, sizeof r->err, "tick %d conference %d: the exchanged mix differs from the whole-conference mix", t, c);
					goto done;
				}
	}
	r->ok = 1;
done:
	if (x) mi_exchange_destroy(x);
	if (mine) mi_mixer_destroy(mine);
	if (whole) mi_mixer_destroy(whole);
	if (ctx) {
		void *dv[] = {d_in, d_out, d_sum, d_all, d_ref};
		for (i = 0; i < 5; ++i)
			if (dv[i]) mi_dev_free(ctx, dv[i]);
		mi_ctx_destroy(ctx);
	}
	free(h_loc), free(h_out), free(h_all), free(h_ref);
	return NULL;
}

int main(int argc, char **argv) {
	int n = argc > 1 ? atoi(argv[1]) : mi_device_count(), k;
	pthread_t th[32];
	static rank_t ranks[32];
	if (n < 1) {
		fprintf(stderr, "no MI355X: %s\n", mi_last_error()); /* there is no CPU fallback */
		return 1;
	}
	while (n > 1 && (MEMBERS % n || n > 32)) --n; /* equal shares of the 32 members */
	if (mi_exchange_unique_id(ranks[0].id, sizeof ranks[0].id) != MI_OK) {
		fprintf(stderr, "exchange: %s\n", mi_last_error());
		return 1;
	}
	for (k = 0; k < n; ++k) {
		ranks[k].rank = k, ranks[k].nranks = n;
		memcpy(ranks[k].id, ranks[0].id, sizeof ranks[0].id);
		pthread_create(&th[k], NULL, rank_main, &ranks[k]);
	}
	for (k = 0; k < n; ++k) pthread_join(th[k], NULL);
	for (k = 0; k < n; ++k)
		if (!ranks[k].ok) {
			fprintf(stderr, "%s\n", ranks[k].err);
			return 1;
		}
	printf("ok %d\n", n);
	return 0;
}
