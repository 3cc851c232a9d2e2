for i in 1 2; do
for V in 4294967295 2048 1024 512 256; do echo "FLAT_MIN=$V"; SPIRAL_FOLD_FLAT_MIN=$V python tools/batch_query.py 1 4; done
done
