"""Alias: the reference's README calls the label CLI ``generate_label_files.py``
(``README.md:59``) while the file is ``gen_label_files.py``; both names work here."""
from .gen_label_files import gen_label_scan, main  # noqa: F401

if __name__ == "__main__":
    main()
