"""CPU oracle for beam search -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``BeamDecoder.forward`` (reference ``src/seq_gen.py:46-242``) and ``get_outputs_until_eos``
(``src/seq_gen.py:6-24``) on top of the oracle model classes.  The decoder is re-run on the whole prefix at every
step exactly like the reference (``:164-166``); no key/value cache.

Deviations from the text of the reference, both forced and both recorded in SURVEY section 8(c):
  * ``:216`` computes ``indices / V`` which is a true division on torch >= 1.5 and cannot index; the restatement
    uses floor division (what torch 1.4, the reference's pinned version, computed).
  * ``torch.topk`` (``:203``) leaves the order among EQUAL scores unspecified, and equal scores are produced on
    purpose (``:194-196`` zeroes whole rows of log-probs).  The restatement pins the order to "lowest flat index
    first" (stable descending sort); the HIP path implements the same rule.

Pinning: ``get_outputs_until_eos`` is checked against the reference's own function (importable in the build
container; vectors in ``tests/golden/beam_kat.json``).  ``BeamDecoder`` results are PARITY UNPINNED by the
reference (it holds no test or fixture for beam search); token ids in ``tests/golden/toy_beam.pt`` come from this
restatement.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def get_outputs_until_eos(eos, outputs, size_limit=None, remove_first_token: bool = False):
    """src/seq_gen.py:6-24: per row, tokens before the first ``eos`` (or up to ``size_limit[r]``)."""
    if outputs.dim() == 1:
        outputs = outputs.unsqueeze(0)
    outputs = outputs.cpu()
    start = 1 if remove_first_token else 0
    result = []
    for r in range(outputs.size(0)):
        hits = (outputs[r] == eos).nonzero()
        if hits.numel() > 0:
            end = int(hits[0, 0])
        else:
            end = outputs.size(1) if size_limit is None else int(size_limit[r])
        result.append(outputs[r, start:end])
    return result


def stable_topk(scores: torch.Tensor, k: int):
    """top-k along dim 1 with ties broken by the lowest index (see module docstring)."""
    vals, idx = torch.sort(scores, dim=1, descending=True, stable=True)
    return vals[:, :k].contiguous(), idx[:, :k].contiguous()


class BeamDecoder(nn.Module):
    """src/seq_gen.py:27-242 (text and image-only branches; the image+text gate mix ``:181-189`` is the branch the
    reference cannot run)."""

    def __init__(self, seq2seq_model, beam_width: int = 5, max_len_a: float = 1.1, max_len_b: int = 5,
                 len_penalty_ratio: float = 0.8):
        super().__init__()
        self.seq2seq_model = seq2seq_model
        self.beam_width = beam_width
        self.max_len_a = max_len_a
        self.max_len_b = max_len_b
        self.len_penalty_ratio = len_penalty_ratio

    def len_penalty(self, lengths: torch.Tensor):
        return torch.pow((lengths + 6.0) / 6.0, self.len_penalty_ratio).unsqueeze(-1)

    @torch.no_grad()
    def forward(self, src_inputs=None, src_sizes=None, first_tokens=None, src_mask=None, src_langs=None,
                tgt_langs=None, pad_idx=None, max_len: int = None, unpad_output: bool = True, beam_width: int = None,
                images=None, proposals=None, image_embed=None, trace: list = None):
        un = lambda x: x[0] if isinstance(x, list) else x
        tgt_langs, first_tokens, src_langs, src_mask = un(tgt_langs), un(first_tokens), un(src_langs), un(src_mask)
        src_sizes, src_inputs, images, image_embed = un(src_sizes), un(src_inputs), un(images), un(image_embed)
        model = self.seq2seq_model
        if beam_width is None:
            beam_width = self.beam_width
        batch_lang = int(tgt_langs[0])
        if src_inputs is not None:
            batch_size = src_inputs.size(0)
        elif images is not None:
            batch_size = images.size(0)
        else:
            batch_size = image_embed.size(0)
        if images is not None and max_len is None:
            max_len = 512

        if src_inputs is not None and images is None:
            src_langs_t = src_langs.unsqueeze(-1).expand(-1, src_inputs.size(-1))
            encoder_states = model.encode(src_inputs, src_mask, src_langs_t)[0]
        elif src_inputs is None:
            encoder_states = model.encode(images=images)[0] if image_embed is None else image_embed
        else:
            raise NotImplementedError("image+text beam search: broken in the reference (SURVEY a16)")
        eos = model.text_processor.sep_token_id()
        V = model.config.vocab_size
        max_pos = model.encoder.embeddings.position_embeddings.num_embeddings
        max_len_func = lambda s: min(int(self.max_len_a * s + self.max_len_b), max_pos)
        if max_len is None:
            max_len = max_len_func(src_inputs.size(1))
        if src_inputs is None:
            max_lens = torch.LongTensor([max_len] * batch_size)
        else:
            max_lens = torch.LongTensor([max_len_func(int(x)) for x in src_sizes])

        outs = first_tokens.unsqueeze(1)            # [B,1] then [B,beam,i]
        scores = torch.zeros(outs.size())
        cur_size = torch.zeros(batch_size) if beam_width > 1 else None
        decoder = model.decoder if not model.lang_dec else model.decoder[batch_lang]
        output_layer = model.output_layer if (not model.lang_dec) and model.tie_embed else model.output_layer[batch_lang]

        for i in range(1, max_len):
            cur = outs.view(-1, outs.size(-1))
            eos_mask = torch.any(cur == eos, 1)
            if int(eos_mask.sum()) == beam_width * batch_size:
                break
            over = (max_lens < (i + 1)).unsqueeze(-1).expand(-1, beam_width)
            rep = 1 if i == 1 else beam_width
            enc = encoder_states if rep == 1 else torch.repeat_interleave(encoder_states, rep, 0)
            langs = tgt_langs.unsqueeze(-1).expand(-1, cur.size(1))
            if rep > 1:
                langs = torch.repeat_interleave(langs, rep, 0)
            enc_mask = None
            if src_inputs is not None:
                enc_mask = src_mask if rep == 1 else torch.repeat_interleave(src_mask, rep, 0)
            states = decoder(encoder_states=enc, input_ids=cur, encoder_attention_mask=enc_mask,
                             tgt_attention_mask=torch.ones(cur.size()), token_type_ids=langs)[:, -1, :]
            lp = F.log_softmax(output_layer(states), dim=-1)
            lp[eos_mask] = 0
            if i > 1:
                lp[over.contiguous().view(-1)] = 0
            total = scores.view(-1).unsqueeze(-1) + lp
            if beam_width > 1:
                total = total / self.len_penalty(cur_size.view(-1))
            top_scores, indices = stable_topk(total.view(batch_size, -1), beam_width)
            if i > 1:
                indices[over] = pad_idx
                flat = indices.view(-1)
                flat[eos_mask] = pad_idx        # NB indexes the NEW slots with the OLD rows' mask (:211-212)
                parent = indices // V           # floor division, see module docstring
                prefix = outs.gather(1, parent.unsqueeze(-1).expand(-1, -1, outs.size(-1))).view(-1, i)
                sizes = cur_size.gather(1, parent).view(-1) if beam_width > 1 else None
            else:
                flat = indices.view(-1)
                prefix = torch.repeat_interleave(outs, beam_width, 0)
                sizes = torch.repeat_interleave(cur_size, beam_width, 0) if beam_width > 1 else None
            word = (flat % V).unsqueeze(-1)
            outs = torch.cat([prefix, word], dim=1).view(batch_size, beam_width, i + 1)
            if beam_width > 1:
                cur_size = (sizes + (word.squeeze(-1) != pad_idx)).view(batch_size, beam_width)
            scores = top_scores
            if trace is not None:
                trace.append({"outs": outs.clone(), "scores": scores.clone(),
                              "sizes": None if cur_size is None else cur_size.clone()})

        best = outs[:, 0, :] if outs.dim() == 3 else outs
        if unpad_output:
            return get_outputs_until_eos(eos, best, size_limit=max_lens)
        best = best.cpu()
        return [best[r] for r in range(best.size(0))]
