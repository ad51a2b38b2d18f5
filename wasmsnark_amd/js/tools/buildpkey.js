#!/usr/bin/env node
/* node buildpkey.js -i proving_key.json -o proving_key.bin   (same options as the reference's tools/buildpkey.js) */
"use strict";
const fs = require("fs");
const { pkeyJsonToBin } = require("../formats.js");
const a = process.argv.slice(2);
const opt = (s, l, d) => { const i = Math.max(a.indexOf(s), a.indexOf(l)); return i >= 0 && i + 1 < a.length ? a[i + 1] : d; };
fs.writeFileSync(opt("-o", "--output", "proving_key.bin"), pkeyJsonToBin(JSON.parse(fs.readFileSync(opt("-i", "--input", "proving_key.json"), "utf8"))));
