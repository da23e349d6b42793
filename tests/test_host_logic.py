"""CPU tests of host-side logic that needs no GPU: the no-code unpickler of the reference's side files, the trainers' flag
table against the reference's command lines, the bench launcher."""
import io
import os
import pickle
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_safe_pickle_reads_reference_files_and_refuses_globals(tmp_path):
    from imagetranslate_amd import safe_pickle
    cfg = (False, False, 6, 6, 512, 2048, False, 1, False)
    p = tmp_path / "mt_config"
    p.write_bytes(pickle.dumps(cfg))
    assert safe_pickle.load_mt_config(str(p)) == cfg
    q = tmp_path / "langs"
    q.write_bytes(pickle.dumps({"<en>": 0, "<fa>": 1}))
    assert safe_pickle.load_langs(str(q)) == {"<en>": 0, "<fa>": 1}

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    p.write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        safe_pickle.load_mt_config(str(p))
    assert not (tmp_path / "pwned").exists()
    p.write_bytes(pickle.dumps((1, 2, 3)))
    with pytest.raises(ValueError):
        safe_pickle.load_mt_config(str(p))
    q.write_bytes(pickle.dumps({"<en>": "x"}))
    with pytest.raises(ValueError):
        safe_pickle.load_langs(str(q))


def test_reference_command_lines_parse():
    """README.md:160-163 (MASS) and :212-216 (MT) of the reference, flag for flag."""
    from imagetranslate_amd.option_parser import get_img_options_parser
    mass = ("--tok sample/tok/ --model sample/mass_model --mass_train sample/en.mass.0,sample/fa.mass.0,sample/ar.mass.0 "
            "--capacity 2800 --batch 16000 --step 300000 --fstep 0 --warmup 100000 --acc 8 --fp16").split()
    o, rest = get_img_options_parser().parse_args(mass)
    assert not rest and o.accum == 8 and o.fp16 and o.total_capacity == 2800 and o.batch == 16000 and o.warmup == 100000
    assert o.mass_train_path.count(",") == 2 and o.finetune_step == 0
    mt = ("--tok sample/tok/ --model sample/mt_model --train_mt sample/fa2en.train.mt --capacity 600 --batch 4000 --beam 4 "
          "--step 500000 --warmup 4000 --fstep 0 --lr 0.0001 --dev_mt sample/fa2en.dev.mt --dropout 0.1 --fp16 "
          "--pretrained sample/mass_model.latest").split()
    o, rest = get_img_options_parser().parse_args(mt)
    assert not rest and o.mt_train_path == "sample/fa2en.train.mt" and o.mt_dev_path == "sample/fa2en.dev.mt"
    assert o.beam_width == 4 and o.learning_rate == 1e-4 and o.pretrained_path == "sample/mass_model.latest"
    # defaults of the reference (src/option_parser.py): --batch 20000, --mask 0.5, --enc 6 --dec 6 --embed 768, 12 heads
    o, _ = get_img_options_parser().parse_args([])
    assert (o.batch, o.mask_prob, o.encoder_layer, o.decoder_layer, o.embed_dim, o.intermediate_layer_dim, o.heads) == (
        20000, 0.5, 6, 6, 768, 3072, 12)
    assert (o.accum, o.mtl_weight, o.clip, o.max_image, o.img_capacity) == (1, 0.1, 1, 32, 50)
    from imagetranslate_amd.caption import get_lm_option_parser
    c, rest = get_lm_option_parser().parse_args("--input imgs --target en --output out.txt --tok tok --model m --beam 4 --fp16".split())
    assert not rest and c.beam_width == 4 and c.batch == 16 and c.target_lang == "en"


def test_bench_self_launch_propagates_child_failure():
    """`python bench.py --gpus 2` without a launcher starts its own ranks; here there is no GPU, so the ranks fail -- the
    parent must report a non-zero exit code (and must not hang)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert "rank" in r.stderr and "exited with code" in r.stderr


def test_unimplemented_flags_are_refused_not_ignored():
    """--lm (train_image_mt), --cont, --save-opt, --dict: parsed like the reference's flags, refused at use (--langs drives the
    back-translation phase since round 3)."""
    from imagetranslate_amd.option_parser import get_img_options_parser
    from imagetranslate_amd.train_image_mt import reject_off_path
    for argv in (["--lm", "x"], ["--cont"], ["--save-opt"], ["--dict", "d"]):
        o, _ = get_img_options_parser().parse_args(argv)
        with pytest.raises(NotImplementedError):
            reject_off_path(o)
    o, _ = get_img_options_parser().parse_args(["--lm", "x"])
    reject_off_path(o, lm_supported=True)  # train_captioning adopts a pretrained MT model through --lm
    o, _ = get_img_options_parser().parse_args([])
    reject_off_path(o)


def test_active_head_hint_is_refused_inside_an_accumulation_window():
    """An idle head is left out of the gradient exchange but scaled by 1/world with the whole buffer: inside a --acc window that
    would shrink a head that was active in an earlier micro-step.  train_step refuses the combination before touching anything."""
    from imagetranslate_amd.parallel import train_step

    class FakeSync:
        world_size = 2
        def begin_step(self, head=None):
            raise AssertionError("the guard must fire first")
    with pytest.raises(ValueError):
        train_step(None, None, {}, sync=FakeSync(), update=False, active_head=0)
    s = FakeSync()
    s._window_open = True   # an earlier micro-step of this window did not update
    with pytest.raises(ValueError):
        train_step(None, None, {}, sync=s, update=True, active_head=1)
