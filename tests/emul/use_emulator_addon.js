// Tests only.  require() this BEFORE wasmsnark_amd/js: the product's index.js then binds the emulator build of the N-API addon
// (tests/emul/wsnark_napi_emul.node -> tests/emul/libwsnark_emul.so: the kernel sources compiled for a CPU thread emulator) in place
// of wasmsnark_amd/js/build/wsnark_napi.node.  Nothing in the product can do this: the addon has no path argument.
const path = require("path");
const Module = require("module");
const product = path.join(__dirname, "..", "..", "wasmsnark_amd", "js", "build", "wsnark_napi.node");
const m = new Module(product, null);
m.filename = product;
m.exports = require(path.join(__dirname, "wsnark_napi_emul.node"));
m.loaded = true;
require.cache[product] = m;
