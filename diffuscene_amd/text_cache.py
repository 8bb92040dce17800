"""Cached text features for text-conditioned training / sampling.

The reference runs the FROZEN ``bert-base-cased`` encoder inside every training step and every ``sample`` call
(diffusion_scene_layout_ddpm.py:47-51, :210-221, :262-273): tokenizer(batch, padding=True) -> BertModel -> last_hidden_state
(B, L, 768) -> the trainable ``fc_text_f``.  The encoder's weights never change, so its output is a function of the description
alone; this cache evaluates it ONCE per distinct description and the training step only runs ``fc_text_f`` + the denoiser.

Equivalence with the reference's per-batch call: with an attention mask, the hidden state of a real token does not depend on
how far the sequence is padded, and the hidden state at a [PAD] position depends only on the real tokens and on that position's
own embedding -- not on the other sequences of the batch.  (The denoiser's cross-attention does not mask padded positions, so
those rows matter.)  Each description is therefore encoded padded to ``max_tokens`` and stored whole; ``batch`` slices every
entry to the longest token count of the batch, which is exactly the tensor ``tokenizer(texts, padding=True)`` + ``BertModel``
produce (tests/test_text_cache.py checks it against a direct batched call of the same encoder).

The encoder itself (weights, tokenizer files) is out of scope of this repository and is passed in by the caller."""
import torch


class BertFeatureCache:
    """description -> last_hidden_state, evaluated once.  ``tokenizer`` / ``model``: a ``transformers`` tokenizer / encoder
    pair (the reference uses BertTokenizer / BertModel 'bert-base-cased'); ``store``: device the cached rows live on."""

    def __init__(self, tokenizer, model, max_tokens=64, store="cpu", encode_device=None):
        self.tokenizer, self.model = tokenizer, model
        self.max_tokens = int(max_tokens)
        self.store = torch.device(store)
        self.encode_device = torch.device(encode_device) if encode_device is not None else None
        self.rows = {}            # text -> (features (max_tokens, H) on self.store, token count)

    def __len__(self):
        return len(self.rows)

    @torch.no_grad()
    def precompute(self, texts, batch_size=64):
        """Encode every description not seen before (deduplicated), ``batch_size`` at a time."""
        todo = [t for t in dict.fromkeys(texts) if t not in self.rows]
        dev = self.encode_device or next(self.model.parameters()).device
        was_training = self.model.training
        self.model.eval()
        for i in range(0, len(todo), batch_size):
            chunk = todo[i:i + batch_size]
            # the reference's tokenizer(text, padding=True) never truncates: a description longer than max_tokens would silently
            # get different features here, so it is an error (build the cache with a larger max_tokens)
            full = self.tokenizer(chunk, padding=False, truncation=False)["input_ids"]
            too_long = [(t, len(ids)) for t, ids in zip(chunk, full) if len(ids) > self.max_tokens]
            if too_long:
                raise ValueError("BertFeatureCache(max_tokens=%d): %d description(s) tokenize longer (longest %d tokens: %r); "
                                 "the reference does not truncate" % (self.max_tokens, len(too_long),
                                                                      max(n for _, n in too_long), too_long[0][0][:80]))
            tok = self.tokenizer(chunk, return_tensors="pt", padding="max_length", truncation=True, max_length=self.max_tokens)
            n_tok = tok["attention_mask"].sum(dim=1).tolist()
            out = self.model(**{k: v.to(dev) for k, v in tok.items()}).last_hidden_state.to(self.store)
            for j, t in enumerate(chunk):
                self.rows[t] = (out[j].contiguous(), int(n_tok[j]))
        if was_training:
            self.model.train()
        return self

    def batch(self, texts, device):
        """(B, L, H) features of ``texts`` with L = the longest token count among them (== tokenizer(texts, padding=True))."""
        self.precompute(texts)
        L = max(self.rows[t][1] for t in texts)
        return torch.stack([self.rows[t][0][:L] for t in texts]).to(device, non_blocking=True)

    def attach_to_samples(self, sample_params, device=None):
        """Add ``desc_bert`` to a batch dict that carries ``description`` (what DiffusionSceneLayout_DDPM consumes when built
        with ``text_bert_cached: true``)."""
        dev = device or next(iter(v for v in sample_params.values() if isinstance(v, torch.Tensor))).device
        sample_params["desc_bert"] = self.batch(sample_params["description"], dev)
        return sample_params

    def state_dict(self):
        return {"max_tokens": self.max_tokens, "rows": {t: (f.cpu(), n) for t, (f, n) in self.rows.items()}}

    def load_state_dict(self, sd):
        if sd["max_tokens"] != self.max_tokens:
            raise ValueError("cache was built with max_tokens=%d, this one uses %d" % (sd["max_tokens"], self.max_tokens))
        self.rows.update({t: (f.to(self.store), n) for t, (f, n) in sd["rows"].items()})
        return self
