"""Synthetic stand-ins for the assets configs/LRS23/AV/EffConfInterCTC.py of the reference loads at import time (SURVEY section 7 "Environment" (b)):

    python tools/make_synthetic_assets.py <root>

writes under <root> (the directory main.py is then run from):
    datasets/LRS3/tokenizerbpe256.model     a 256-piece sentencepiece BPE model trained on synthetic text                (AV cfg :41,64)
    datasets/LRS3/6gram_lrs23.arpa          an empty n-gram stub (beam search falls back to greedy: ctcdecode/KenLM absent) (AV cfg :42)
    callbacks/LRW/EffConfCE/checkpoints_epoch_30_step_57247.ckpt   an LRW-shaped checkpoint: the state_dict of a seeded VisualEfficientConformerCE,
                                                                    whose `encoder.front_end.*` entries the config transplants      (AV cfg :27,70-75)
The licensed corpora stay absent: nnet.datasets.LRS then yields synthetic LRS2-shaped clips."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(root):
    import torch
    import nnet
    os.makedirs(os.path.join(root, "datasets", "LRS3"), exist_ok=True)
    os.makedirs(os.path.join(root, "callbacks", "LRW", "EffConfCE"), exist_ok=True)
    tok = os.path.join(root, "datasets", "LRS3", "tokenizerbpe256")
    if not os.path.exists(tok + ".model"):
        import sentencepiece as spm
        rnd = random.Random(0)
        syll = ["ka", "to", "mi", "re", "su", "no", "ha", "li", "ve", "do", "an", "er", "ing", "th", "st", "qu", "ow", "ea", "ly", "ch"]
        words = ["".join(rnd.choice(syll) for _ in range(rnd.randint(1, 4))).upper() for _ in range(600)]
        txt = tok + ".txt"
        with open(txt, "w") as f:
            for _ in range(4000):
                f.write(" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 12))) + "\n")
        spm.SentencePieceTrainer.train(input=txt, model_prefix=tok, vocab_size=256, model_type="bpe", character_coverage=1.0, bos_id=-1, eos_id=-1, unk_id=1, pad_id=0,
                                       minloglevel=2)
        os.remove(txt)
    with open(os.path.join(root, "datasets", "LRS3", "6gram_lrs23.arpa"), "w") as f:
        f.write("\\data\\\nngram 1=0\n\n\\1-grams:\n\n\\end\\\n")
    ck = os.path.join(root, "callbacks", "LRW", "EffConfCE", "checkpoints_epoch_30_step_57247.ckpt")
    if not os.path.exists(ck):
        torch.manual_seed(30)
        lrw = nnet.VisualEfficientConformerCE(vocab_size=500)
        torch.save({"model_state_dict": lrw.state_dict(), "model_step": 57247, "is_distributed": False}, ck)
    print("synthetic assets under", os.path.abspath(root))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
