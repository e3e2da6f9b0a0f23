"""oracle/ — CPU restatement (plain PyTorch fp32) of the reference's one-step restoration path.

TEST INFRASTRUCTURE ONLY. Nothing here is shipped or measured as the product: only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this package, and only as the checker. The product path
(instarevive_amd/) never imports it and fails loudly when the HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):
  * SwinIR, LDM-VAE Encoder/Decoder, PixArtMS block wiring, eps_to_mu, _sliding_windows, colour fix, auto_resize/pad
    are pinned against outputs of the reference's own modules imported from /root/reference in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * The DiT / VAE arithmetic the CLI actually executes lives in diffusers==0.30.0 (requirements.txt:2), which is
    neither in the tree nor installed: for the diffusers-only behaviours (3-D encoder_attention_mask used as an
    additive bias, PatchEmbed pos-emb regeneration, VAE state-dict key names) parity is UNPINNED; they are restated
    from the published diffusers 0.30.0 algorithm and anchored on the reference's call sites
    (test_scripts/inference.py:106-117, scripts/DMD/transformer_train/generate.py:54-87) and its converter key map
    (tools/convert_pixart_to_diffusers.py:30-180).
"""
