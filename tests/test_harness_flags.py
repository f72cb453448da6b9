"""CPU side of the flag-compatible harness (tools/run_path.py): the reference's example command lines parse to the values
the reference's own parser gives them (flag names, short forms and defaults are the drop-in contract: utils.py:7-83)."""
from tools import run_path

# the argument lists of the example commands in the reference's README.md (lines 99-186), script names dropped
README_COMMANDS = [
    ("-nf 2000 -m pretrained_models/saved_model_5PC_l_epi/model.net -bs 32 -fmat 0 -sam 1 -bm 1 -t 2 -pth data",
     dict(nfeatures=2000, batch_size=32, fmat=0, sampler=1, batch_mode=1, threshold=2.0, data_path="data")),
    ("-nf 2000 -tr 1 -bs 1 -lr 0.000001 -t 3. -sam 3 -fmat 1 -sid loftr -m2 diff_ransac_models/loftr_model.pth -pth data",
     dict(tr=1, batch_size=1, learning_rate=1e-6, threshold=3.0, sampler=3, fmat=1, session="loftr")),
    ("-nf 2000 -m w.net -bs 32 -fmat 0 -sam 2 -tr 1 -w2 1 -t 0.75 -pth data",
     dict(fmat=0, sampler=2, tr=1, w2=1.0, threshold=0.75, model="w.net")),
    ("-nf 2000 -m w.net -bs 32 -fmat 1 -sam 3 -tr 1 -w2 1 -t 0.75 -pth data", dict(fmat=1, sampler=3, tr=1)),
    ("-nf 2000 -sam 2 -tr 1 -t 0.75 -pth data", dict(sampler=2, tr=1, ransac_batch_size=64, batch_size=32)),
    ("-nf 2000 -tr 1 -bs 1 -lr 1e-6 -t 0.75 -sam 3 -fmat 1 -w2 1 -sid loftr -e 50 -p 0 -topk 1 -m2 x.ckpt -pth data/",
     dict(epochs=50, prob=0, topk="1", model_loftr="x.ckpt")),
    ("-nf 2000 -m pretrained_models/saved_model_5PC_l_epi/model.net -bs 32 -fmat 1 -sam 3 -ds sacre_coeur -t 2 -pth data",
     dict(datasets="sacre_coeur", threshold=2.0, fmat=1, sampler=3)),
]


def test_reference_command_lines_parse():
    for argv, expect in README_COMMANDS:
        opt = run_path.parse(argv.split())
        for k, v in expect.items():
            assert getattr(opt, k) == v, (argv, k, getattr(opt, k), v)
        assert opt.ignored == []


def test_defaults_are_the_references():
    opt = run_path.parse([])
    assert (opt.device, opt.nfeatures, opt.batch_size, opt.ransac_batch_size, opt.fmat, opt.scoring, opt.sampler,
            opt.precision, opt.tr, opt.threshold, opt.weighted, opt.prob, opt.k, opt.snn) == \
        ("cuda", 2000, 32, 64, 0, 1, 1, 1, 0, 0.75, 0, 2, 300, 0.80)


def test_foreign_script_arguments_are_reported_not_fatal():
    opt = run_path.parse("-d cpu -us 0 -pth x".split())     # test_magsac_point.py's extra flag
    assert opt.device == "cpu" and opt.ignored == ["-us", "0"]
