"""Decoders (nnet/decoders.py).  Greedy CTC: device argmax (HIP) + exact integer collapse; beam search with KenLM/GPT rescoring depends
on the un-vendored `ctcdecode` C++ package and is out of scope."""
import torch
import torch.nn as nn

from .. import ops


def ctc_collapse(tokens, length, blank=0):
    """unique_consecutive then drop blanks (nnet/decoders.py:105-113) -- bit-exact index work"""
    out, prev = [], None
    for t in tokens[:length]:
        if t != prev and t != blank:
            out.append(t)
        prev = t
    return out


class CTCGreedySearchDecoder(nn.Module):
    def __init__(self, tokenizer_path=None, blank_token=0):
        super().__init__()
        self.blank_token = blank_token
        self.tokenizer = None
        if tokenizer_path is not None:
            import os
            if os.path.exists(tokenizer_path):
                import sentencepiece as spm
                self.tokenizer = spm.SentencePieceProcessor(tokenizer_path)

    def token_ids(self, logits, logits_len):
        am = ops.argmax_rows(logits).cpu().tolist()
        lens = logits_len.cpu().tolist()
        return [ctc_collapse(seq, int(n), self.blank_token) for seq, n in zip(am, lens)]

    def forward(self, outputs, from_logits=True):
        if from_logits:
            ids = self.token_ids(outputs[0], outputs[1])
        else:
            tokens, lens = outputs
            ids = [t[:int(n)].tolist() for t, n in zip(tokens.cpu(), lens.cpu())]
        return self.tokenizer.decode(ids) if self.tokenizer is not None else ids


class CTCBeamSearchDecoder(CTCGreedySearchDecoder):
    """Constructor signature of nnet/decoders.py:134 kept so that configs import; decoding falls back LOUDLY to greedy search."""

    def __init__(self, tokenizer_path=None, beam_size=16, ngram_path=None, ngram_tmp=1.0, ngram_alpha=0.6, ngram_beta=1.0, ngram_offset=100,
                 neural_config_path=None, neural_checkpoint=None, neural_alpha=0.6, neural_beta=1.0, num_processes=8, test_time_aug=False):
        super().__init__(tokenizer_path=tokenizer_path)
        import warnings
        warnings.warn("CTCBeamSearchDecoder: ctcdecode/KenLM are not available -- greedy CTC search is used (beam search is out of scope, SURVEY row 23)")
        self.beam_size = beam_size


decoder_dict = {"CTCGreedySearchDecoder": CTCGreedySearchDecoder, "CTCBeamSearchDecoder": CTCBeamSearchDecoder}
