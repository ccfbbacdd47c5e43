python __graft_entry__.py smoke 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
