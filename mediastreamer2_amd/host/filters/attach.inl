// filters/attach.inl -- what is decided when a graph is attached, on the ATTACHING thread.
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, last); not compiled on its own.
//
// ms_ticker_attach runs preprocess() of every filter of the graph on the thread that attaches, one after the other, BEFORE the graph's
// sources join the ticker's execution list (src/base/msticker.c:153-183; f->ticker is set right in front of each call, msfilter.c:266-270).
// The reference's filters do all their set-up there -- speex_echo_state_init, the mixer's tick buffers -- and the ticker thread's
// first tick finds them ready.  Up to round 5 this plugin decided what to fuse, opened the fused banks and reset their slots at the
// first block of every leg, ON THE TICKER THREAD: the first ticks of a loaded ticker took 0.2 - 2.7 s.
//
// Every facade's preprocess ends in graph_preprocessed().  The LAST facade of the graph to get there finds every facade's ticker
// set, and does the graph's fusing: conferences (sending legs or a server's remote members), legs without a mixer (an AudioStream's
// sending side, its encoder included), receiving sides (decoder -> PLC -> flow control).  The first tick then stages and launches
// like any other.  What is looked at are THIS plugin's facades only: somebody else's filter that is preprocessed later changes nothing.
// (The facades' first process() still looks -- fuse_checked / fuse_state -- for a graph whose facades were configured after the
// attach; normally it finds the decision made.)

// Is every facade of this plugin in f's graph attached to f's ticker already -- is f the last of them to be preprocessed?  No lock: the graph
// is being attached by THIS thread (ms_ticker_attach sets a filter's ticker right in front of its preprocess, msfilter.c:266-270) and nobody
// walks it yet.  The facades that are not the last leave their preprocess without looking for their hub: with sixteen threads attaching 2 048
// streams of seventeen filters each, the look-ups in the process-wide registry were two thirds of the attach (profiles/r06_attach_profile.txt).
bool graph_ready(MSFilter *f) {
	if (!f->ticker) return false;
	std::vector<MSFilter *> todo{f};
	std::unordered_set<MSFilter *> seen{f};
	while (!todo.empty()) {
		MSFilter *g = todo.back();
		todo.pop_back();
		if (is_ours(g->desc) && g->ticker != f->ticker) return false;
		for (int i = 0; i < g->desc->ninputs; ++i)
			if (g->inputs[i] && g->inputs[i]->prev.filter && seen.insert(g->inputs[i]->prev.filter).second) todo.push_back(g->inputs[i]->prev.filter);
		for (int i = 0; i < g->desc->noutputs; ++i)
			if (g->outputs[i] && g->outputs[i]->next.filter && seen.insert(g->outputs[i]->next.filter).second) todo.push_back(g->outputs[i]->next.filter);
	}
	return true;
}

void graph_preprocessed(MSFilter *f) { // (hub locked by the caller)
	if (!f->ticker) return;
	std::vector<MSFilter *> todo{f}, all;
	std::unordered_set<MSFilter *> seen{f};
	while (!todo.empty()) {
		MSFilter *g = todo.back();
		todo.pop_back();
		if (is_ours(g->desc)) {
			if (g->ticker != f->ticker) return; // a facade of the graph is still to be preprocessed: the last one does the work
			all.push_back(g);
		}
		for (int i = 0; i < g->desc->ninputs; ++i)
			if (g->inputs[i] && g->inputs[i]->prev.filter && seen.insert(g->inputs[i]->prev.filter).second) todo.push_back(g->inputs[i]->prev.filter);
		for (int i = 0; i < g->desc->noutputs; ++i)
			if (g->outputs[i] && g->outputs[i]->next.filter && seen.insert(g->outputs[i]->next.filter).second) todo.push_back(g->outputs[i]->next.filter);
	}
	// conferences first (a mixer in conference mode: its legs, or a server's remote members, conf_try_fuse looks at both shapes)
	for (MSFilter *g : all)
		if (g->desc == &ms_mi355x_audio_mixer_desc && ((MixerState *)g->data)->conf_mode != 0) conf_try_fuse(g);
	// sending legs without a conference mixer: headed by MSResample (through a mic_equalizer) or by MSSpeexEC itself
	for (MSFilter *g : all) {
		if (!is_ec_desc(g->desc)) continue;
		SpeexECState *s = (SpeexECState *)g->data;
		if (s->leg || !s->configured || s->unsupported || __atomic_load_n(&s->bypass_live, __ATOMIC_RELAXED) || !g->inputs[1]) continue;
		MSFilter *rs = g->inputs[1]->prev.filter;
		if (rs && rs->desc == &ms_mi355x_equalizer_desc) rs = rs->inputs[0] ? rs->inputs[0]->prev.filter : NULL;
		if (rs && rs->desc == &ms_mi355x_resample_desc) {
			ResampleData *rd = (ResampleData *)rs->data;
			if (rd->input_rate == rd->output_rate || rd->leg || rd->pool) continue; // (a forwarder, or a filter that has run on its facade: its first block looks, as before)
			rd->fuse_checked = true;
			if (MSFilter *mx = leg_find_mixer(rs)) conf_try_fuse(mx);
			else leg_try_fuse_plain(rs);
			continue;
		}
		s->fuse_checked = true;
		if (MSFilter *mx = leg_find_mixer_ec(g)) conf_try_fuse(mx);
		else leg_try_fuse_plain_ec(g);
	}
	// receiving sides: decoder -> [local_mixer] -> MSGenericPLC -> [MSAudioFlowControl]
	for (MSFilter *g : all)
		if (is_g711_dec(g->desc) || g->desc == &ms_mi355x_generic_plc_desc) recv_chain_preprocessed(g);
	// ... and every facade that did NOT join a batch takes the slot of its own it will need, here: a bank that has to be opened (pinned and
	// device memory, the batch objects) is opened by the attaching thread, not by the ticker's first ticks.  (A mixer that can only forward
	// never mixes, a forwarding MSResample never resamples: no slot.  MSResample takes its slots with its first block: one per channel.)
	for (MSFilter *g : all) {
		if (g->desc == &ms_mi355x_volume_desc) {
			VolumeData *d = (VolumeData *)g->data;
			if (!d->leg && !d->sleg && !d->meter_leg && !d->pool) volume_attach_slot(g);
		} else if (is_ec_desc(g->desc)) {
			SpeexECState *s = (SpeexECState *)g->data;
			if (!s->leg && !__atomic_load_n(&s->bypass_live, __ATOMIC_RELAXED)) ec_acquire(g);
		} else if (g->desc == &ms_mi355x_audio_mixer_desc) {
			MixerState *s = (MixerState *)g->data;
			if (!s->fbank && !s->sbank && !s->one_input) mixer_acquire(g);
		} else if (is_g711_dec(g->desc)) {
			MapFilter *d = (MapFilter *)g->data;
			if (!d->rleg && !d->sleg) map_attach(g, d, d->law ? OP_ULAW_DEC : OP_ALAW_DEC);
		} else if (is_g711_enc(g->desc)) {
			MapFilter *d = (MapFilter *)g->data;
			if (!d->fleg && !d->sleg) map_attach(g, d, d->law ? OP_ULAW_ENC : OP_ALAW_ENC);
		} else if (g->desc == &ms_mi355x_generic_plc_desc) {
			PlcFilter *d = (PlcFilter *)g->data;
			if (!d->rleg && d->rate > 0) plc_attach(g, d);
		} else if (g->desc == &ms_mi355x_equalizer_desc) {
			EqualizerData *d = (EqualizerData *)g->data;
			if (!d->leg && d->active) equalizer_attach(g); // (an inactive one forwards: no slot until it is switched on.  Here and not in its own preprocess: a mic_equalizer that
			                                               // joins a leg would take a slot of its own first -- bank, history read-back, release: a second per 2 048 legs)
		} else if (g->desc == &ms_mi355x_audio_flow_control_desc) {
			FlowFilter *d = (FlowFilter *)g->data;
			if (!d->rleg) {
				const bool had = d->pool != nullptr;
				if (flowctl_attach(g, d) && had) MI_MUST(mi_flowctl_reset(d->pool->fc, d->slot, 1)); // flowcontrol.c:166-169 (a new slot is at rest already)
			}
		}
	}
}
void generic_preprocess(MSFilter *f) { // a facade with nothing of its own to prepare
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f);
}

// does this facade's work run in a device-resident batch shared with its neighbours (a fused sending leg, a conference, a server's member, a
// stream's receiving side) rather than in a bank of its own?  (hub locked)
bool facade_in_batch(MSFilter *f) {
	const MSFilterDesc *d = f->desc;
	if (d == &ms_mi355x_resample_desc) return ((ResampleData *)f->data)->leg != nullptr;
	if (is_ec_desc(d)) return ((SpeexECState *)f->data)->leg != nullptr;
	if (d == &ms_mi355x_volume_desc) return ((VolumeData *)f->data)->leg || ((VolumeData *)f->data)->sleg || ((VolumeData *)f->data)->meter_leg;
	if (d == &ms_mi355x_equalizer_desc) return ((EqualizerData *)f->data)->leg != nullptr;
	if (d == &ms_mi355x_audio_mixer_desc) return ((MixerState *)f->data)->fbank || ((MixerState *)f->data)->sbank;
	if (is_g711_dec(d) || is_g711_enc(d)) return ((MapFilter *)f->data)->sleg || ((MapFilter *)f->data)->rleg || ((MapFilter *)f->data)->fleg;
	if (d == &ms_mi355x_generic_plc_desc) return ((PlcFilter *)f->data)->rleg != nullptr;
	if (d == &ms_mi355x_audio_flow_control_desc) return ((FlowFilter *)f->data)->rleg != nullptr;
	return false;
}
