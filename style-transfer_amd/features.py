"""Feature-list layout of the precomputed DDIM-inversion features and their merge.

A feature file (`precomputed_feats_k/<name>_cnt.pkl` / `<name>_sty.pkl`) is a list of 50 dicts,
index i <-> timestep 981 - 20*i, each with 'config', 'z_enc' and 18 tensors
`output_block_{6..11}_self_attn_{q,k,v}` of shape (8, N, d) (generate_pkl_only.py:101-142).
"""

NUM_STEPS = 50


def feat_merge_2sty(opt, cnt_feats, sty_feats_1, sty_feats_2, start_step=0):
    """Re-key three feature lists into the per-step dicts the UNet consumes (run_ori.py:36-75).

    For every step i >= 50 - start_step and every key K of the content dict:
      K ending in 'q'        -> out[K] = content;  out[K+'_sty1'] = char;  out[K+'_sty2'] = back
      K ending in 'k' / 'v'  -> out[K] = char;     out[K+'_cnt'] = content; out[K+'_sty2'] = back
    Tensors are referenced, never copied.  Every step dict carries
    'config' = {'gamma', 'T', 'timestep': i, 'cnt_k': None, 'sty_q': None}.
    A style list that holds no queries (a packed side-car keeps only what the attention reads: nothing
    downstream reads '..._q_sty1' / '..._q_sty2') simply yields no such entries.
    """
    merged = [{'config': {'gamma': opt.gamma, 'T': opt.T, 'timestep': i, 'cnt_k': None, 'sty_q': None}}
              for i in range(NUM_STEPS)]
    for i in range(NUM_STEPS):
        if i < (NUM_STEPS - start_step):
            continue
        cnt, sty1, sty2 = cnt_feats[i], sty_feats_1[i], sty_feats_2[i]
        step = merged[i]
        for key in cnt.keys():
            is_q = key.endswith('q')
            is_kv = key.endswith('k') or key.endswith('v')
            if is_q:
                step[key] = cnt[key]
                if key in sty1:
                    step[key + '_sty1'] = sty1[key]
            if is_kv:
                step[key] = sty1[key]
                step[key + '_cnt'] = cnt[key]
            if is_kv or (is_q and key in sty2):
                step[key + '_sty2'] = sty2[key]
    return merged


def feat_merge(opt, cnt_feats, sty_feats, start_step=0):
    """The single-style merge (hf_run.py:40-68), for every step i >= 50 - start_step and every key K of the
    style dict:  K ending in 'q' -> out[K] = content, out[K+'_sty'] = style;  K ending in 'k' / 'v' ->
    out[K] = style, out[K+'_cnt'] = content.  The attention then runs the no-mask branch
    (ldm/modules/attention.py:569-579): blended query, style keys and values, temperature T."""
    merged = [{'config': {'gamma': opt.gamma, 'T': opt.T, 'timestep': i, 'cnt_k': None, 'sty_q': None}}
              for i in range(NUM_STEPS)]
    for i in range(NUM_STEPS):
        if i < (NUM_STEPS - start_step):
            continue
        cnt, sty = cnt_feats[i], sty_feats[i]
        step = merged[i]
        for key in sty.keys():
            if key[-1] == 'q':
                step[key] = cnt[key]
                step[key + '_sty'] = sty[key]
            if key[-1] == 'k' or key[-1] == 'v':
                step[key] = sty[key]
                step[key + '_cnt'] = cnt[key]
    return merged


def adain(cnt_feat, sty_feat):
    """((cnt - mean_c) / std_c) * std_s + mean_s per channel over (batch, H, W), unbiased std
    (run_ori.py:77-82), in one kernel."""
    from . import ops
    return ops.masked_adain(cnt_feat.contiguous(), sty_feat.contiguous())


def masked_adain_init(cnt_z_enc, char_z_enc, back_z_enc, mask):
    """Initial latent of the two-style sampler (run_ori.py:228-232): the raw [0,1] content mask
    (H0, W0), resized bilinearly to the latent, blends adain(cnt, char) and adain(cnt, back)."""
    import torch
    from . import ops
    m = torch.as_tensor(mask, dtype=torch.float32, device=cnt_z_enc.device).contiguous()
    return ops.masked_adain(cnt_z_enc.contiguous(), char_z_enc.contiguous(), back_z_enc.contiguous(), m)
